// catan_abi.hip - host side of libcatan_hip.so: handle, launches, C ABI (include/catan_hip.h).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/catan_hip.h"
#include "../../include/catan_hip_nn.h"
#include "../../include/catan_hip_tuning.h"
#include "catan_kernels.hip"
#include "catan_obs.hip"
#include "catan_ppo.hip"
#include "catan_nn.hip"
#include "catan_tile_encoder.hip"
#include "catan_heads.hip"
#include "catan_collector.hip"
#include "catan_rows.hip"
#include "catan_te_bwd.hip"
#include "catan_optim.hip"
#include "catan_wgrad_big.hip"

using namespace catan;

struct catan_env {
    int device;
    long n, N;
    Ctx ctx;
    catan_cfg_t cfg;
    void* state;          // W rows then B rows
    u32* mpk;             // packed masks [N][16]
    u32* spec_state;      // shadow records / masks for speculative re-deals inside a lock-step step (enqueue_slow)
    u32* spec_mpk;
    u32 spec_epoch;       // lock-step step counter: tags the shadows of a step
    int ctr_clean;        // lock-step: the counters are maintained by the kernels themselves (no memset per step) once set
    int lock_parity;      // ... and the sort alternates between its two count sets
    u32* err;             // invalid-action counter
    double* reward64;     // caller-owned [n][4] unrounded rewards of catan_step (catan_set_reward_f64_buffer), or NULL
    // scratch for catan_random_rollout
    i32* scratch_actions; // [n][18]
    float* scratch_reward;// [n][4]
    u8* scratch_done;     // [n]
    Pending pend;         // tier-2 longest-road hand-off (device arrays)
    unsigned long long* prof; // device [12] phase profile of k_step, enabled by catan_profile_enable
    int prof_on;
    int lr_mid_budget;    // deferred windows: the middle tier's budget (0: off), and k_lr_heavy's workgroups behind it
    int lr_mid_heavy_grid;
    int t1_group;         // the library's own deferred loop: passes per tier-1 launch (1, or 2: deferred_iter_grouped)
    int g_open, g_slot, g_passes; int64_t g_count;   // ... its running group
    int lr_split;         // tier 1 as search (k_lr_finish<LRF_SPLIT>) + lane-per-game completion (k_lr_complete)
    int step_bin_order;   // k_step: longest-lasting bins first (StepCfg::bin_order)
    int step_wpb;         // waves per k_step workgroup (4: one workgroup per CU, a SIMD per wave; 1: one-wave workgroups)
    u32* prof_wave;       // [N/64][8] per-wave phase ticks of the last k_step (catan_profile_enable(env, 2))
    u32* pctr;            // [N] per-game decision counters of the random policy (deferred rollouts)
    int lr_budget[3];     // tier-1 longest-road iteration budget: [0] lock-step, [1] deferred (tails are amortised there), [2] the fused-sampling loop
    int lr_round[2];      // tier-2 iterations per bulk-synchronous round: [0] lock-step, [1] deferred
    int deferred_fused;   // catan_set_deferred_fused (CATAN_DEFERRED_FUSED at creation)
    int fused_subs;       // ... its sub-lists per bin (CATAN_FUSED_SUBS at creation: 1, 2, 4, 8 or 16; Pending::nsub while that loop runs)
    int step_games;       // games per k_step wave: 64, 32 or 16 (catan_set_step_wave_games; CATAN_STEP_WAVE_GAMES at creation)
    hipStream_t side;     // re-deals run here, concurrently with the longest-road kernels on the caller's stream
    hipEvent_t ev_fork, ev_join;
    hipStream_t fstream[2];  // deferred rollouts: tier-1 longest road + completion of iteration t run on fstream[t & 1] during t+1
    hipEvent_t ev_fready[3], ev_fdone[3];   // per tier-1 slot (the library's own deferred loop rotates three: deferred_iter_legacy)
    float* f_reward;      // [n][4] / [n]: outputs of the completions that run on fstream (scratch)
    u8* f_done;
    hipStream_t sstream;  // deferred rollouts: tier 2 + re-deals of window w run here during window w+1
    hipEvent_t ev_sdone[2];
    float* s_reward;      // outputs of the completions that run on sstream (scratch)
    u8* s_done;
    // catan_step_deferred / catan_step_flush: the deferred schedule for caller-supplied actions
    int64_t d_it;         // calls since the last flush (0: no deferred sequence is open)
    int d_window;         // its window length
    hipStream_t d_stream; // ... and the caller's stream (one stream per sequence)
    MtPair* mt_dev;       // RNG contract (A): the handle's two MT19937 generators on the device (catan_seed_mt19937), else NULL
};

// games per k_step wave (catan_set_step_wave_games).  32 since the end of round 5: two waves per SIMD, so that one wave's gather overlaps the other's
// rules code - 41.4 -> 40.9 us per pass, lock-step 180.8 -> 178 us (profiles/r05_s5_pass_experiments.txt, run 20; 16: 43.2 us).  In round 3 the
// same switch bought nothing (28.3 -> 27.6 us of k_step): the launch was then bound by the CUs tier 2 held, not by its waves.
constexpr int DEFAULT_STEP_WAVE_GAMES = 32;
// sub-lists per sort bin in the fused-sampling deferred loop (Pending::bctr): one counter per bin serialises the launch's range reservations
constexpr int DEFAULT_FUSED_SUBS = 8;
constexpr int LR_BUDGET_DEFERRED = 12;   // (swept together with the window length: tools/deferred_sweep.py)
// The fused-sampling loop (round 6; profiles/r06_t1_stagger_ab.txt): tier 1's launch is STAGGERED - a one-wave kernel that idles T1_STAGGER_US in front of
// it on the side stream - so that the next pass's k_step, which becomes eligible at the same moment, gets its workgroups dispatched first (with both
// kernels' workgroups interleaved, tier 1's 3 072-4 096 one-wave workgroups held the LDS and registers k_step's second wave per SIMD needs: its waves
// started over ~20 us, 9 % of them in a second round; staggered: within 5-7 us, 2 %).  Any delay from 1 to 6 us gives the same 38.9 -> 36.9 us per pass:
// it is the ORDER of dispatch that matters, not the time.  With it, 4 096 tier-1 workgroups and budget 10 measure best (36.0 us).
constexpr int T1_STAGGER_US = 4, LR_GRID_FUSED = 4096, LR_BUDGET_FUSED = 10;
// cross-stream ordering inside one device: no timing, no system-scope release (which would flush L2 at every record)
constexpr unsigned EV_SYNC = hipEventDisableTiming | hipEventDisableSystemFence;
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(x)                                                                                    \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) return fail(CATAN_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <class T>
static int attn_dispatch(bool bwd, const void* qkv, const int* lens, const void* dout, void* out, long B, int L, int H, int HD, hipStream_t st) {
    const T* q = (const T*)qkv;
    // bf16: the MFMA kernels (one wave per sequence); fp32 and unaligned buffers: the VALU kernels
    const bool mfma = sizeof(T) == 2 && (((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)dout) & 15) == 0;
    const unsigned nb4 = (unsigned)((B + 3) / 4);
    const unsigned short* q16 = (const unsigned short*)qkv;
    if (L == 19 && H == 4 && HD == 16) {
        unsigned nb = (unsigned)((B + 2) / 3);
        if (mfma && !bwd) hipLaunchKernelGGL((k_attn_mfma_fwd<19, 4, 16>), dim3(nb4), dim3(256), 0, st, q16, lens, (unsigned short*)out, B);
        else if (mfma && lens) hipLaunchKernelGGL((k_attn_mfma_bwd<19, 4, 16, true>), dim3(nb4), dim3(256), 0, st, q16, lens, (const unsigned short*)dout, (unsigned short*)out, B);
        else if (mfma) hipLaunchKernelGGL((k_attn_mfma_bwd<19, 4, 16, false>), dim3(nb4), dim3(256), 0, st, q16, lens, (const unsigned short*)dout, (unsigned short*)out, B);
        else if (!bwd) hipLaunchKernelGGL((k_attn_fwd<T, 19, 4, 16>), dim3(nb), dim3(64), 0, st, q, lens, (T*)out, B);
        else hipLaunchKernelGGL((k_attn_bwd<T, 19, 4, 16>), dim3(nb), dim3(64), 0, st, q, lens, (const T*)dout, (T*)out, B);
    } else if (L == 25 && H == 4 && HD == 4) {
        unsigned nb = (unsigned)((B + 1) / 2);
        if (mfma && !bwd) hipLaunchKernelGGL((k_attn_mfma_fwd<25, 4, 4>), dim3(nb4), dim3(256), 0, st, q16, lens, (unsigned short*)out, B);
        else if (mfma && lens) hipLaunchKernelGGL((k_attn_mfma_bwd<25, 4, 4, true>), dim3(nb4), dim3(256), 0, st, q16, lens, (const unsigned short*)dout, (unsigned short*)out, B);
        else if (mfma) hipLaunchKernelGGL((k_attn_mfma_bwd<25, 4, 4, false>), dim3(nb4), dim3(256), 0, st, q16, lens, (const unsigned short*)dout, (unsigned short*)out, B);
        else if (!bwd) hipLaunchKernelGGL((k_attn_fwd<T, 25, 4, 4>), dim3(nb), dim3(64), 0, st, q, lens, (T*)out, B);
        else hipLaunchKernelGGL((k_attn_bwd<T, 25, 4, 4>), dim3(nb), dim3(64), 0, st, q, lens, (const T*)dout, (T*)out, B);
    } else return fail(CATAN_EINVAL, "catan_attention: unsupported (L, heads, head_dim); built for (19,4,16) and (25,4,4)");
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
template <class T, int D, int GL>
static int ln_launch(bool bwd, const void* x, const float* w, const float* b, const void* dy, void* out, float* dw, float* db,
                     long rows, float eps, int relu, hipStream_t st) {
    long nb = (rows + (256 / GL) - 1) / (256 / GL);
    const long cap = bwd ? 2048 : 8192;          // bwd ends with 2*D atomics per block: keep the grid moderate
    if (nb > cap) nb = cap;
    if (!bwd) hipLaunchKernelGGL((k_ln_fwd<T, D, GL>), dim3((unsigned)nb), dim3(256), 0, st, (const T*)x, w, b, (T*)out, rows, eps, relu);
    else hipLaunchKernelGGL((k_ln_bwd<T, D, GL>), dim3((unsigned)nb), dim3(256), 0, st, (const T*)x, w, b, (const T*)dy, (T*)out, dw, db, rows, eps, relu);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
template <class T, int EPL, int GL>
static int lnw_launch(bool bwd, const void* x, const float* w, const float* b, const void* dy, void* out, float* dw, float* db,
                      long rows, float eps, int relu, hipStream_t st) {
    constexpr int RPB = 256 / GL;
    long nb = (rows + RPB - 1) / RPB;
    const long cap = bwd ? 1024 : 8192;          // bwd ends with 2*D atomics per block
    if (nb > cap) nb = cap;
    if (!bwd) hipLaunchKernelGGL((k_lnw_fwd<T, EPL, GL>), dim3((unsigned)nb), dim3(256), 0, st, (const T*)x, w, b, (T*)out, rows, eps, relu);
    else hipLaunchKernelGGL((k_lnw_bwd<T, EPL, GL>), dim3((unsigned)nb), dim3(256), 0, st, (const T*)x, w, b, (const T*)dy, (T*)out, dw, db, rows, eps, relu);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
template <class T>
static int ln_dispatch(bool bwd, const void* x, const float* w, const float* b, const void* dy, void* out, float* dw, float* db,
                       long rows, int D, float eps, int relu, hipStream_t st) {
    switch (D) {
    case 16: return ln_launch<T, 16, 16>(bwd, x, w, b, dy, out, dw, db, rows, eps, relu, st);
    case 25:
        if constexpr (sizeof(T) == 2) {      // bf16, enough rows, 16-byte aligned tensors: a lane per row through LDS (k_lnr_*)
            if (rows >= 4096 && (((uintptr_t)x | (uintptr_t)out | (uintptr_t)dy) & 15) == 0) {
                long nb = (rows + 255) / 256;
                if (nb > 2048) nb = 2048;
                if (!bwd) hipLaunchKernelGGL((k_lnr_fwd<25>), dim3((unsigned)nb), dim3(256), 0, st, (const unsigned short*)x, w, b, (unsigned short*)out, rows, eps, relu);
                else hipLaunchKernelGGL((k_lnr_bwd<25>), dim3((unsigned)nb), dim3(256), 0, st, (const unsigned short*)x, w, b, (const unsigned short*)dy,
                                        (unsigned short*)out, dw, db, rows, eps, relu);
                HIPCHK(hipGetLastError());
                return CATAN_OK;
            }
        }
        return ln_launch<T, 25, 32>(bwd, x, w, b, dy, out, dw, db, rows, eps, relu, st);
    case 32: return ln_launch<T, 32, 32>(bwd, x, w, b, dy, out, dw, db, rows, eps, relu, st);
    case 64:
        if ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)dy) & 15) == 0) return lnw_launch<T, 8, 8>(bwd, x, w, b, dy, out, dw, db, rows, eps, relu, st);
        return ln_launch<T, 64, 64>(bwd, x, w, b, dy, out, dw, db, rows, eps, relu, st);
    // 128 / 256 wide: 8 elements per lane (16-byte accesses for bf16), 16 / 32 lanes per row - with one wave per row a lane moved 4 or 8
    // bytes per access and the kernels ran at 1.6 TB/s (204 800 x 128: 100 us backward; 16-byte lanes: see profiles/r03_*)
    case 128:
        if ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)dy) & 15) == 0) return lnw_launch<T, 8, 16>(bwd, x, w, b, dy, out, dw, db, rows, eps, relu, st);
        return lnw_launch<T, 2, 64>(bwd, x, w, b, dy, out, dw, db, rows, eps, relu, st);
    case 256:
        if ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)dy) & 15) == 0) return lnw_launch<T, 8, 32>(bwd, x, w, b, dy, out, dw, db, rows, eps, relu, st);
        return lnw_launch<T, 4, 64>(bwd, x, w, b, dy, out, dw, db, rows, eps, relu, st);
    case 512: return lnw_launch<T, 8, 64>(bwd, x, w, b, dy, out, dw, db, rows, eps, relu, st);
    default: return fail(CATAN_EINVAL, "catan_layer_norm: unsupported width (built for 16, 25, 32, 64, 128, 256, 512)");
    }
}

template <class T>
static int lstm_cell_launch(bool bwd, const void* gx, const void* gh, const float* c_prev, const float* mask, const float* dh, const float* dc,
                            void* o0, float* o1, long n, int L, hipStream_t st) {
    long nb = (n * (L / 4) + 255) / 256;
    if (nb > 16384) nb = 16384;
    if (!bwd) hipLaunchKernelGGL((k_lstm_cell_fwd<T>), dim3((unsigned)nb), dim3(256), 0, st, (const T*)gx, (const T*)gh, c_prev, mask, (float*)o0, o1, n, L);
    else hipLaunchKernelGGL((k_lstm_cell_bwd<T>), dim3((unsigned)nb), dim3(256), 0, st, (const T*)gx, (const T*)gh, c_prev, mask, dh, dc, (T*)o0, o1, n, L);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

// the column slice [col0, col0 + W) of a layer whose input is `ld` wide (W, col0, ld multiples of 8; W + 1 <= 160): X rows are strided
// How the rows of a weight gradient are split over the grid.  Every block ends with one atomic per element of its O x (I + 1)
// accumulator tile, so the grid is a balance between streaming parallelism and atomics (measured on MI355X, tools/bench_wgrad_shapes.py:
// 614 400 x (160 -> 256): 456 us with up to 1 024 blocks, 307 us with 256; 3.4 M x (64 -> 128): 340 / 288 / 395 us with 1 024 / 512 /
// 256; 3.4 M x (64 -> 32): 158 / 184 / 290 us): the wider the tile, the fewer blocks; long inputs take 16 stages per block, short ones 8.
static void wgrad_grid(long R, int I, int O, long& nb, long& per) {
    const long stages = (R + WG_KT - 1) / WG_KT;
    const long tile = (long)((O + 15) / 16 * 16) * ((I + 1 + 15) / 16 * 16);
    const long cap = tile >= 24576 ? 256 : (tile >= 3072 ? 512 : 1024);
    const long spb = R >= 131072 ? 16 : 8;
    nb = stages / spb < 1 ? 1 : (stages / spb < cap ? stages / spb : cap);
    per = (stages + nb - 1) / nb * WG_KT;
    nb = (R + per - 1) / per;
}
template <int OTW>
static int wgrad_launch_slice(const void* x, const void* dy, float* dw, float* db, long R, int ld, int col0, int W, int O, hipStream_t st) {
    long nb, per;
    wgrad_grid(R, W, O, nb, per);
    hipLaunchKernelGGL((k_wgrad_tr<OTW, 10>), dim3((unsigned)nb), dim3(256), 0, st, (const unsigned short*)x, (const unsigned short*)dy, dw, db, R, W, O, per,
                       (long)ld, col0);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
template <int OTW, int IT>
static int wgrad_launch(const void* x, const void* dy, float* dw, float* db, long R, int I, int O, hipStream_t st) {
    long nb, per;
    wgrad_grid(R, I, O, nb, per);
    if ((I & 7) == 0 && (O & 7) == 0)      // 16 B vectors never straddle a row: row-major LDS image + transposing LDS reads
        hipLaunchKernelGGL((k_wgrad_tr<OTW, IT>), dim3((unsigned)nb), dim3(256), 0, st, (const unsigned short*)x, (const unsigned short*)dy, dw, db, R, I, O, per);
    else
        hipLaunchKernelGGL((k_wgrad<OTW, IT>), dim3((unsigned)nb), dim3(256), 0, st, (const unsigned short*)x, (const unsigned short*)dy, dw, db, R, I, O, per);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
template <int OTW>
static int wgrad_dispatch_it(int it, const void* x, const void* dy, float* dw, float* db, long R, int I, int O, hipStream_t st) {
    if (it <= 2) return wgrad_launch<OTW, 2>(x, dy, dw, db, R, I, O, st);
    if (it <= 5) return wgrad_launch<OTW, 5>(x, dy, dw, db, R, I, O, st);
    if (it <= 10) return wgrad_launch<OTW, 10>(x, dy, dw, db, R, I, O, st);
    return fail(CATAN_EINVAL, "catan_linear_wgrad: in_features + 1 > 160");
}

template <int OTW, int IT>
static int wgrad_launch_grouped(bool tr, const WgBatch& b, long blocks, hipStream_t st) {
    if (tr) hipLaunchKernelGGL((k_wgrad_tr_grouped<OTW, IT>), dim3((unsigned)blocks), dim3(256), 0, st, b);
    else hipLaunchKernelGGL((k_wgrad_grouped<OTW, IT>), dim3((unsigned)blocks), dim3(256), 0, st, b);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
static int wgrad_launch_grouped_dispatch(bool tr, int otw, int itc, const WgBatch& b, long blocks, hipStream_t st) {
#define CATAN_WG_CASE(A, B) if (otw == A && itc == B) return wgrad_launch_grouped<A, B>(tr, b, blocks, st)
    CATAN_WG_CASE(1, 2); CATAN_WG_CASE(1, 5); CATAN_WG_CASE(1, 10);
    CATAN_WG_CASE(2, 2); CATAN_WG_CASE(2, 5); CATAN_WG_CASE(2, 10);
    CATAN_WG_CASE(3, 2); CATAN_WG_CASE(3, 5); CATAN_WG_CASE(3, 10);
    CATAN_WG_CASE(4, 2); CATAN_WG_CASE(4, 5); CATAN_WG_CASE(4, 10);
#undef CATAN_WG_CASE
    return fail(CATAN_EINVAL, "catan_linear_wgrad_grouped: no kernel for this tile shape");
}

template <int RT, int WAVES>
static int head_launch_cfg(const HeadArgs& a, hipStream_t st) {
    const dim3 grid((unsigned)((a.B + WAVES * RT * 16 - 1) / (WAVES * RT * 16))), block(WAVES * 64);
    switch ((a.K + 15) / 16) {
    case 1: hipLaunchKernelGGL((k_head_fwd<1, RT, WAVES>), grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL((k_head_fwd<2, RT, WAVES>), grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL((k_head_fwd<3, RT, WAVES>), grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL((k_head_fwd<4, RT, WAVES>), grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL((k_head_fwd<5, RT, WAVES>), grid, block, 0, st, a); break;
    }
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
static int head_launch(const HeadArgs& a, hipStream_t st) {
    return a.B <= HD_NARROW_MAX_ROWS ? head_launch_cfg<HD_RT_NARROW, HD_WAVES_NARROW>(a, st) : head_launch_cfg<HD_RT_WIDE, HD_WAVES_WIDE>(a, st);
}

template <int KS>
static int linrows_dispatch_nt(int nt, const void* x, const void* w, const void* b, void* y, long R, int K, int N, const void* aux, int mode, hipStream_t st) {
    long nb = ((R + 15) / 16 + 3) / 4;
    if (nb > 2048) nb = 2048;
#define CATAN_LINROWS(NT) hipLaunchKernelGGL((k_linear_rows<KS, NT>), dim3((unsigned)nb), dim3(256), 0, st, (const unsigned short*)x, \
                                             (const unsigned short*)w, (const unsigned short*)b, (unsigned short*)y, R, K, N, (const unsigned short*)aux, mode)
    if (nt <= 1) CATAN_LINROWS(1);
    else if (nt <= 2) CATAN_LINROWS(2);
    else if (nt <= 4) CATAN_LINROWS(4);
    else if (nt <= 8) CATAN_LINROWS(8);
    else CATAN_LINROWS(12);
#undef CATAN_LINROWS
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

// the wide-row backward with the gradient of a second use of x added in (dx = LayerNorm'(dy) + dres): D = 64 (16-byte aligned rows), 128, 256, 512
template <class T>
static int lnw_res_dispatch(const void* x, const float* w, const float* b, const void* dy, const void* dres, void* dx, float* dw, float* db,
                            long rows, int D, float eps, int relu, hipStream_t st) {
#define CATAN_LNW_RES(EPL, GL) do { constexpr int RPB = 256 / GL; long nb = (rows + RPB - 1) / RPB; if (nb > 1024) nb = 1024; \
        hipLaunchKernelGGL((k_lnw_bwd<T, EPL, GL>), dim3((unsigned)nb), dim3(256), 0, st, (const T*)x, w, b, (const T*)dy, (T*)dx, dw, db, rows, eps, relu, (const T*)dres); } while (0)
    switch (D) {
    case 64: CATAN_LNW_RES(8, 8); break;
    case 128: if ((((uintptr_t)x | (uintptr_t)dx | (uintptr_t)dy | (uintptr_t)dres) & 15) == 0) CATAN_LNW_RES(8, 16); else CATAN_LNW_RES(2, 64); break;
    case 256: if ((((uintptr_t)x | (uintptr_t)dx | (uintptr_t)dy | (uintptr_t)dres) & 15) == 0) CATAN_LNW_RES(8, 32); else CATAN_LNW_RES(4, 64); break;
    case 512: CATAN_LNW_RES(8, 64); break;
    default: return fail(CATAN_EINVAL, "catan_layer_norm_bwd_res: built for the widths 64, 128, 256, 512");
    }
#undef CATAN_LNW_RES
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

extern "C" {

const char* catan_last_error(void) { return g_err.c_str(); }

#ifndef CATAN_BUILD_HASH_STR
#define CATAN_BUILD_HASH_STR "unhashed-build-0000000000000000"
#endif
// "CATAN_BUILD_HASH=" + the sha256 prefix of csrc/* and include/catan_hip.h the build script computed (_lib.source_hash):
// the marker makes it readable from the file without loading the library
static const char g_build_hash[] = "CATAN_BUILD_HASH=" CATAN_BUILD_HASH_STR;
const char* catan_build_hash(void) { return g_build_hash + 17; }

void catan_cfg_default(catan_cfg_t* c) {
    c->max_proposed_trades_per_turn = 4; c->win_reward = 500.0; c->dense_reward = 0;
    c->reward_annealing_factor = 1.0; c->validate_actions = 1; c->auto_reset = 1; c->max_actions_per_turn = -1;
}
static inline Limits limits_of(const catan_env_t* e) { return Limits{ e->cfg.max_proposed_trades_per_turn, e->cfg.max_actions_per_turn }; }
int32_t catan_state_words(void) { return STATE_WORDS; }
int32_t catan_mask_words(void) { return MASK_BITS; }
int32_t catan_action_words(void) { return ACTION_WORDS; }
int32_t catan_obs_floats(void) { return OBS_FLOATS; }
int32_t catan_state_bytes_per_game(void) { return STATE_BYTES_PER_GAME; }
int64_t catan_num_envs(const catan_env_t* e) { return e ? e->n : 0; }

static inline hipStream_t S(catan_stream_t s) { return (hipStream_t)s; }
static inline unsigned blocks(long n, int b) { return (unsigned)((n + b - 1) / b); }

static int launch_masks(catan_env_t* e, hipStream_t st) {
    hipLaunchKernelGGL(k_masks, dim3(blocks(e->N, BLOCK)), dim3(BLOCK), 0, st, e->ctx, e->mpk, limits_of(e));
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

int catan_create(catan_env_t** out, int device, int64_t n_envs, uint64_t seed, uint64_t env_id0, const catan_cfg_t* cfg) {
    if (!out || n_envs <= 0) return fail(CATAN_EINVAL, "catan_create: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(CATAN_ENODEV, "catan_create: no HIP device (the HIP path has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(CATAN_EINVAL, "catan_create: bad device index");
    HIPCHK(hipSetDevice(device));
    catan_env* e = new (std::nothrow) catan_env();
    if (!e) return fail(CATAN_ENOMEM, "catan_create: host allocation failed");
    memset(e, 0, sizeof *e);
    e->device = device;
    e->n = n_envs;
    e->N = (n_envs + BLOCK - 1) / BLOCK * BLOCK;
    if (cfg) e->cfg = *cfg; else catan_cfg_default(&e->cfg);
    size_t bytes = (size_t)e->N * STATE_BYTES_PER_GAME;
    hipError_t rc = hipMalloc(&e->state, bytes);
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->mpk, (size_t)e->N * MPK_STRIDE * sizeof(u32));
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->spec_state, (size_t)e->N * REC * sizeof(u32));
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->spec_mpk, (size_t)e->N * MPK_STRIDE * sizeof(u32));
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->err, 64);
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->scratch_actions, (size_t)e->n * ACTION_WORDS * sizeof(i32));
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->scratch_reward, (size_t)e->n * 4 * sizeof(float));
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->scratch_done, (size_t)e->n);
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->pend.ctr, CTR_WORDS * sizeof(u32));
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->pend.req[0], (size_t)e->N * sizeof(u64));
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->pend.req[1], (size_t)e->N * sizeof(u64));
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->pend.req[2], (size_t)e->N * sizeof(u64));
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->f_reward, (size_t)e->n * 4 * sizeof(float));
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->f_done, (size_t)e->n);
    // HIP multiplexes streams onto 4 hardware queues; streams that share a queue serialise.  Caller's stream + side +
    // fstream + sstream = 4, so the two tier-1 slots share one stream (tier 1 of an iteration must fit in one iteration).
    // (A second tier-1 stream was measured with GPU_MAX_HW_QUEUES=4 and 8: 883 M and 447 M env-steps/s against 1 015 M -
    // two tier-1 launches in flight take the SIMDs from k_step.)
    // CATAN_LR_CUS=k (diagnostics, tools/pass_experiments.py): the tier-1 stream confined to k of the CUs (every (cus / k)-th bit of the
    // CU mask), so that its one-wave workgroups do not hold LDS on the CUs k_step's 29 KB workgroups need
    int lr_cus = getenv("CATAN_LR_CUS") ? atoi(getenv("CATAN_LR_CUS")) : 0;
    if (rc == hipSuccess && lr_cus > 0) {
        hipDeviceProp_t prop; int dev = 0;
        int cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        if (lr_cus > cus) lr_cus = cus;
        std::vector<uint32_t> mask((cus + 31) / 32, 0u);
        const char* pat = getenv("CATAN_LR_CUS_PATTERN");             // "block": the first k CUs; default: evenly spread
        for (int k = 0; k < lr_cus; k++) { const int b = (pat && pat[0] == 'b') ? k : (int)((long)k * cus / lr_cus); mask[b >> 5] |= 1u << (b & 31); }
        rc = hipExtStreamCreateWithCUMask(&e->fstream[0], (uint32_t)mask.size(), mask.data());
    } else if (rc == hipSuccess) rc = hipStreamCreateWithFlags(&e->fstream[0], hipStreamNonBlocking);
    e->fstream[1] = e->fstream[0];
    for (int i = 0; i < 3 && rc == hipSuccess; i++) {
        rc = hipEventCreateWithFlags(&e->ev_fready[i], EV_SYNC);
        if (rc == hipSuccess) rc = hipEventCreateWithFlags(&e->ev_fdone[i], EV_SYNC);
    }
    for (int i = 0; i < 2 && rc == hipSuccess; i++) {
        rc = hipMalloc((void**)&e->pend.heavy[i], (size_t)e->N * sizeof(u64));
        if (rc == hipSuccess) rc = hipMalloc((void**)&e->pend.resets[i][0], (size_t)e->N * sizeof(i32));
        if (rc == hipSuccess) rc = hipMalloc((void**)&e->pend.resets[i][1], (size_t)e->N * sizeof(i32));
        if (rc == hipSuccess) rc = hipMalloc((void**)&e->pend.resets[i][2], (size_t)e->N * sizeof(i32));
        if (rc == hipSuccess) rc = hipEventCreateWithFlags(&e->ev_sdone[i], EV_SYNC);
    }
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->pend.heavy2, (size_t)e->N * sizeof(u64));
    if (rc == hipSuccess) rc = hipStreamCreateWithFlags(&e->sstream, hipStreamNonBlocking);
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->s_reward, (size_t)e->n * 4 * sizeof(float));
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->s_done, (size_t)e->n);
    // the sort's lists: BIN_SETS sets x NBINS bins x (fused-sampling rollouts) fused_subs sub-lists of up to N game ids each - a game is in at most
    // one list, so no sub-list can overflow whatever the split; only what is used is ever touched (65 536 games, 8 sub-lists: 151 MB of address space)
    e->fused_subs = DEFAULT_FUSED_SUBS;
    if (const char* fs = getenv("CATAN_FUSED_SUBS")) { const int v = atoi(fs); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) e->fused_subs = v; }
    static_assert(DEFAULT_FUSED_SUBS <= MAX_SUBS, "sub-lists per bin");
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->pend.lists, (size_t)BIN_SETS * NBINS * e->fused_subs * e->N * sizeof(i32));
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->pend.bctr, (size_t)BIN_SETS * NBINS * MAX_SUBS * BCTR_PAD * sizeof(u32));
    if (rc == hipSuccess) rc = hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking);
    if (rc == hipSuccess) rc = hipEventCreateWithFlags(&e->ev_fork, EV_SYNC);
    if (rc == hipSuccess) rc = hipEventCreateWithFlags(&e->ev_join, EV_SYNC);
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->pend.type, (size_t)e->N);
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->pend.who, (size_t)e->N);
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->pend.len, (size_t)e->N * sizeof(u64));
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->pend.arrive, (size_t)e->N * sizeof(u32));
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->pend.spec, (size_t)e->N * sizeof(u64));
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->pend.busy, (size_t)e->N);
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->pctr, (size_t)e->N * sizeof(u32));
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->prof_wave, (size_t)prof_wave_rows(e->N) * 8 * sizeof(u32));
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->prof, PROF_WORDS * sizeof(unsigned long long));
    if (rc != hipSuccess) { catan_destroy(e); return fail(CATAN_ENOMEM, std::string("catan_create: hipMalloc: ") + hipGetErrorString(rc)); }
    HIPCHK(hipMemset(e->state, 0, bytes));
    HIPCHK(hipMemset(e->err, 0, 64));
    HIPCHK(hipMemset(e->spec_mpk, 0, (size_t)e->N * MPK_STRIDE * sizeof(u32)));   // no shadow carries the tag of a step yet (epochs start at 1)
    HIPCHK(hipMemset(e->pend.ctr, 0, CTR_WORDS * sizeof(u32)));
    HIPCHK(hipMemset(e->pend.type, 0, (size_t)e->N));
    HIPCHK(hipMemset(e->pend.busy, 0, (size_t)e->N));
    HIPCHK(hipMemset(e->pend.len, 0, (size_t)e->N * sizeof(u64)));      // (bit 63 of a game's word is k_lr_finish<LRF_SPLIT>'s mark for k_lr_complete)
    HIPCHK(hipMemset(e->pctr, 0, (size_t)e->N * sizeof(u32)));
    e->lr_budget[0] = LR_BUDGET; e->lr_budget[1] = LR_BUDGET_DEFERRED; e->lr_budget[2] = LR_BUDGET_FUSED;
    e->lr_round[0] = LR_ROUND_LOCKSTEP; e->lr_round[1] = LR_ROUND;
    e->step_games = DEFAULT_STEP_WAVE_GAMES;
    // the fused-sampling loop is the library's own deferred loop since round 6 (one kernel per pass on the main stream; with per-bin sub-lists,
    // 32-game waves and the middle tier: 38.9 us per pass against 41.4 for sampler + k_step, profiles/r06_fused_loop_ab.txt); CATAN_DEFERRED_FUSED=0: the sampler form
    e->deferred_fused = 1;
    if (const char* df = getenv("CATAN_DEFERRED_FUSED")) e->deferred_fused = atoi(df) != 0;
    // one-wave workgroups by default.  Four waves per workgroup (CATAN_STEP_WAVES_PER_BLOCK=4: one 116 KB workgroup per CU, a SIMD per wave) was
    // measured SLOWER (k_step 31.8 -> 39.6 us, pass 54.4 -> 61.8 us, profiles/r05_k_step_pass_experiments.txt): the tier-1 waves of the
    // previous pass hold LDS on most CUs, so a 116 KB workgroup often has to wait for a CU where the 29 KB one-wave workgroup fits at once
    e->step_wpb = 1;
    if (const char* wp = getenv("CATAN_STEP_WAVES_PER_BLOCK")) e->step_wpb = atoi(wp) == 4 ? 4 : 1;
    e->step_bin_order = 1;   // on since round 5 (54.5 -> 52.3-53.4 us per pass: profiles/r05_s5_pass_experiments.txt); CATAN_STEP_BIN_ORDER=0: bins in index order
    if (const char* bo = getenv("CATAN_STEP_BIN_ORDER")) e->step_bin_order = atoi(bo) != 0;
    // tier 1 as search + lane-per-game completion: in the library's own deferred loop since round 5 (a tier-1 launch there has two passes to finish and its
    // waves share the SIMDs with the sampler and k_step: 44.7 -> 43.8 us per pass), not inside a lock-step step (one more kernel on its critical path:
    // 184 -> 197 us) nor in catan_step_deferred (a launch per call: 64.0 -> 65.3 us per call).
    // CATAN_LR_SPLIT=0: never, 2: everywhere
    e->lr_split = 1;
    e->t1_group = 2;       // on since round 5 (47.8 -> 45.5 us per pass at 88.9 instead of 90.0 % active games: +3.6 % env-steps/s, profiles/r05_s5_pass_experiments.txt);
    if (const char* tg = getenv("CATAN_T1_GROUP")) e->t1_group = atoi(tg) == 1 ? 1 : 2;   // CATAN_T1_GROUP=1: a tier-1 launch per pass (deferred_iter_legacy, CATAN_T1_DEPTH slots)
    // the middle tier of a deferred window: on since round 5 (budget 256, 32 tier-2 workgroups behind it: 53.9 -> 51.2-51.6 us per pass,
    // same file); CATAN_LR_MID_BUDGET=0: every tier-2 request straight to k_lr_heavy on 128 workgroups
    e->lr_mid_budget = 256; e->lr_mid_heavy_grid = 32;
    if (const char* mb = getenv("CATAN_LR_MID_BUDGET")) { e->lr_mid_budget = atoi(mb) > 0 ? atoi(mb) : 0; if (e->lr_mid_budget == 0) e->lr_mid_heavy_grid = 128; }
    if (const char* mg = getenv("CATAN_LR_MID_HEAVY_GRID")) { const int g = atoi(mg); if (g >= 8 && g <= 256) e->lr_mid_heavy_grid = g; }
    if (const char* ls = getenv("CATAN_LR_SPLIT")) e->lr_split = atoi(ls) == 0 ? 0 : (atoi(ls) == 2 ? 2 : 1);
    if (const char* sg = getenv("CATAN_STEP_WAVE_GAMES")) { const int g = atoi(sg); if (g == 64 || g == 32 || g == 16) e->step_games = g; }
    e->pend.fa = 0; e->pend.ftag = 1; e->pend.sa = 0; e->pend.stag = 1; e->pend.sample = 0; e->pend.bnext = 0; e->pend.brel = -1; e->pend.bclear = 1; e->pend.lrq_clear = -1;
    e->pend.nsub = 1;
    HIPCHK(hipMemset(e->pend.bctr, 0, (size_t)BIN_SETS * NBINS * MAX_SUBS * BCTR_PAD * sizeof(u32)));
    HIPCHK(hipMemset(e->mpk, 0, (size_t)e->N * MPK_STRIDE * sizeof(u32)));
    e->ctx.R = (u32*)e->state;
    e->ctx.N = e->N; e->ctx.n = e->n;
    e->ctx.key0 = (u32)seed; e->ctx.key1 = (u32)(seed >> 32);
    e->ctx.env_id0 = env_id0;
    int r = catan_reset(e, nullptr, nullptr);
    if (r != CATAN_OK) { catan_destroy(e); return r; }
    HIPCHK(hipStreamSynchronize(nullptr));
    *out = e;
    return CATAN_OK;
}

void catan_destroy(catan_env_t* e) {
    if (!e) return;
    hipSetDevice(e->device);
    if (e->state) hipFree(e->state);
    if (e->mpk) hipFree(e->mpk);
    if (e->spec_state) hipFree(e->spec_state);
    if (e->spec_mpk) hipFree(e->spec_mpk);
    if (e->err) hipFree(e->err);
    if (e->scratch_actions) hipFree(e->scratch_actions);
    if (e->scratch_reward) hipFree(e->scratch_reward);
    if (e->scratch_done) hipFree(e->scratch_done);
    if (e->prof) hipFree(e->prof);
    if (e->pend.ctr) hipFree(e->pend.ctr);
    if (e->pend.req[0]) hipFree(e->pend.req[0]);
    if (e->pend.req[1]) hipFree(e->pend.req[1]);
    if (e->pend.req[2]) hipFree(e->pend.req[2]);
    if (e->f_reward) hipFree(e->f_reward);
    if (e->f_done) hipFree(e->f_done);
    if (e->fstream[0]) hipStreamDestroy(e->fstream[0]);
    for (int i = 0; i < 3; i++) { if (e->ev_fready[i]) hipEventDestroy(e->ev_fready[i]); if (e->ev_fdone[i]) hipEventDestroy(e->ev_fdone[i]); }
    for (int i = 0; i < 2; i++) {
        if (e->pend.heavy[i]) hipFree(e->pend.heavy[i]);
        if (e->pend.resets[i][0]) hipFree(e->pend.resets[i][0]);
        if (e->pend.resets[i][1]) hipFree(e->pend.resets[i][1]);
        if (e->pend.resets[i][2]) hipFree(e->pend.resets[i][2]);
        if (e->ev_sdone[i]) hipEventDestroy(e->ev_sdone[i]);
    }
    if (e->pend.heavy2) hipFree(e->pend.heavy2);
    if (e->sstream) hipStreamDestroy(e->sstream);
    if (e->s_reward) hipFree(e->s_reward);
    if (e->s_done) hipFree(e->s_done);
    if (e->pend.lists) hipFree(e->pend.lists);
    if (e->pend.bctr) hipFree(e->pend.bctr);
    if (e->mt_dev) hipFree(e->mt_dev);
    if (e->side) hipStreamDestroy(e->side);
    if (e->ev_fork) hipEventDestroy(e->ev_fork);
    if (e->ev_join) hipEventDestroy(e->ev_join);
    if (e->pend.type) hipFree(e->pend.type);
    if (e->pend.who) hipFree(e->pend.who);
    if (e->pend.len) hipFree(e->pend.len);
    if (e->pend.arrive) hipFree(e->pend.arrive);
    if (e->pend.spec) hipFree(e->pend.spec);
    if (e->pend.busy) hipFree(e->pend.busy);
    if (e->pctr) hipFree(e->pctr);
    if (e->prof_wave) hipFree(e->prof_wave);
    delete e;
}

static int deferred_open_error(const catan_env_t* e, const char* what);
#define NOT_DEFERRED(e, what) do { int r_ = deferred_open_error(e, what); if (r_ != CATAN_OK) return r_; } while (0)
// contract (A): the generators' output rings topped up in front of a kernel that may draw
static int mt_refill(catan_env_t* e, hipStream_t st) {
    if (e->ctx.mt == nullptr) return CATAN_OK;
    hipLaunchKernelGGL(k_mt_refill, dim3(1), dim3(64), 0, st, e->ctx);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
#define NOT_MT(e, what) do { if ((e)->ctx.mt != nullptr) return fail(CATAN_EINVAL, std::string(what) + ": not available for a handle under the MT19937 contract (catan_seed_mt19937)"); } while (0)
int catan_reset(catan_env_t* e, const uint8_t* reset_mask, catan_stream_t stream) {
    if (!e) return fail(CATAN_EINVAL, "catan_reset: null handle");
    NOT_DEFERRED(e, "catan_reset");
    int r = mt_refill(e, S(stream));
    if (r != CATAN_OK) return r;
    hipLaunchKernelGGL(k_reset, dim3((unsigned)(e->N < 16384 ? e->N : 16384)), dim3(64), 0, S(stream), e->ctx, reset_mask, 0);
    HIPCHK(hipGetLastError());
    return launch_masks(e, S(stream));
}
int catan_reset_board_only(catan_env_t* e, catan_stream_t stream) {
    if (!e) return fail(CATAN_EINVAL, "catan_reset_board_only: null handle");
    NOT_DEFERRED(e, "catan_reset_board_only");
    int r = mt_refill(e, S(stream));
    if (r != CATAN_OK) return r;
    hipLaunchKernelGGL(k_reset, dim3((unsigned)(e->N < 16384 ? e->N : 16384)), dim3(64), 0, S(stream), e->ctx, (const u8*)nullptr, 1);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

// ---- RNG contract (A): MT19937 as numpy (init_genrand) and CPython (init_by_array with a one-word key) seed it; the outputs are generated on the
// device (k_mt_refill), here only the 624-word start states
static void mt_host_init_genrand(uint32_t* mt, uint32_t s) {
    mt[0] = s;
    for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
}
static void mt_host_init_by_array(uint32_t* mt, const uint32_t* key, int klen) {
    mt_host_init_genrand(mt, 19650218u);
    int i = 1, j = 0;
    for (int k = 624 > klen ? 624 : klen; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= 624) { mt[0] = mt[623]; i = 1; }
        if (j >= klen) j = 0;
    }
    for (int k = 623; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= 624) { mt[0] = mt[623]; i = 1; }
    }
    mt[0] = 0x80000000u;
}
// installs generator `which` (its 624 state words and position) with an empty output ring; the consumers' positions restart at 0
static int mt_install(catan_env_t* e, int which, const uint32_t* key, int pos, hipStream_t st) {
    if (e->n != 1) return fail(CATAN_EINVAL, "MT19937 contract: a single-game handle only (the reference's generators are process-global: SURVEY.md 8.4)");
    if (e->d_it > 0) return fail(CATAN_EINVAL, "MT19937 contract: a deferred step sequence is open");
    if (!e->mt_dev) {
        HIPCHK(hipMalloc((void**)&e->mt_dev, sizeof(MtPair)));
        HIPCHK(hipMemset(e->mt_dev, 0, sizeof(MtPair)));
        // (an uninstalled generator: position 624 of an all-zero state; both are installed by catan_seed_mt19937)
    }
    HIPCHK(hipStreamSynchronize(st));
    MtGen* g = which == 0 ? &e->mt_dev->np : &e->mt_dev->py;
    uint32_t head[626];
    memcpy(head, key, 624 * sizeof(uint32_t));
    head[624] = (uint32_t)pos; head[625] = 0u;                 // idx, produced
    HIPCHK(hipMemcpy(g, head, sizeof head, hipMemcpyHostToDevice));
    if (which == 0) HIPCHK(hipMemset(e->ctx.R + W_RNG, 0, sizeof(u32)));           // the numpy stream's position = the game's draw counter
    else HIPCHK(hipMemset(&e->mt_dev->cons_py, 0, sizeof(u32)));
    e->ctx.mt = e->mt_dev;
    return CATAN_OK;
}
int catan_seed_mt19937(catan_env_t* e, uint32_t numpy_seed, uint32_t python_seed, catan_stream_t stream) {
    if (!e) return fail(CATAN_EINVAL, "catan_seed_mt19937: null handle");
    uint32_t mt[624];
    mt_host_init_genrand(mt, numpy_seed);                      // np.random.seed(int)
    int r = mt_install(e, 0, mt, 624, S(stream));
    if (r != CATAN_OK) return r;
    mt_host_init_by_array(mt, &python_seed, 1);                // random.seed(int), int < 2**32
    return mt_install(e, 1, mt, 624, S(stream));
}
int catan_mt19937_set_state(catan_env_t* e, int32_t which, const uint32_t* key624, int32_t pos, catan_stream_t stream) {
    if (!e || !key624 || (which != 0 && which != 1) || pos < 0 || pos > 624) return fail(CATAN_EINVAL, "catan_mt19937_set_state: bad arguments");
    return mt_install(e, which, key624, pos, S(stream));
}

static StepCfg step_cfg(const catan_env_t* e) {
    StepCfg sc;
    sc.validate = e->cfg.validate_actions; sc.dense_reward = e->cfg.dense_reward; sc.win_reward = e->cfg.win_reward;
    sc.annealing = e->cfg.reward_annealing_factor; sc.lim = limits_of(e); sc.auto_reset = e->cfg.auto_reset;
    sc.reward64 = e->reward64;
    sc.prof = e->prof_on ? e->prof : nullptr;
    sc.prof_wave = e->prof_on >= 2 ? e->prof_wave : nullptr;
    sc.prof_timeline = e->prof_on == 3;
    sc.bin_order = e->step_bin_order;
    return sc;
}
// One env step = the games listed by action type (k_sample_random / k_classify), then k_step (fused: apply + done/reward +
// next masks for every game that needs no longest-road update) - the FAST path.  The few games that placed a road /
// settlement or ended take the SLOW path: k_lr_finish (tier-1 path search + completion, one game per wave), k_lr_heavy
// (tier 2 + completion), k_reset_list (re-deal, one wave per finished game).  Its launch times are the latency tails of a
// handful of serial searches / re-deals.
//   lock-step (catan_step, catan_random_rollout): the slow path runs inside every step; the re-deals (real ones and
//     speculative successors of the games the step may still end) run on a side stream right behind k_step.
//   deferred (catan_random_rollout_deferred): a game on the slow path is BUSY (no action, no policy draw).  Tier 1 of
//     iteration t runs on a side stream during iteration t+1 and its games play again at t+2; tier 2 and the re-deals of
//     window w (W iterations) run on another stream during window w+1 and their games play again in window w+2.  Every
//     game's own trajectory is unchanged because its policy stream is indexed by its own decision counter; the release
//     points are fixed by this schedule (the main stream waits for the side work there), never by kernel timing.
constexpr int LR_HEAVY_GRID = 256;   // one 1024-thread workgroup per CU; requests x split parts are strided over them
constexpr int LR_HEAVY_GRID_DEFERRED = 128;  // next to the fast path: leave most CUs (and their LDS) to k_step
constexpr int LR_GRID = 4096;
constexpr int LR_GRID_DEFERRED = 3072;
constexpr int LR_MID_GRID = 2048;          // the middle tier: one request per one-wave workgroup (a window leaves ~640)
constexpr int LR_COMPLETE_GRID = 128;   // k_lr_complete: 64 requests per one-wave workgroup, grid-stride beyond 8 192 requests
constexpr int RESET_GRID = 2048;
// ev (optional): [0] before the sort, [1] after it, [2] after k_step
static int enqueue_fast(catan_env_t* e, const int32_t* actions, float* reward, uint8_t* done, hipStream_t st, hipEvent_t* ev,
                        bool have_hist = false) {
    StepCfg sc = step_cfg(e);
    if (ev) HIPCHK(hipEventRecord(ev[0], st));
    u32* bins = e->pend.ctr + 16 + NBINS * e->pend.bsel;
    if (!have_hist) {                                     // (the rollout loops sort inside k_sample_random)
        hipLaunchKernelGGL(k_classify, dim3(blocks(e->N, BLOCK)), dim3(BLOCK), 0, st, e->ctx, actions, bins, e->pend.lists + (size_t)e->pend.bsel * NBINS * e->N,
                           e->pend.ctr + 4, 12, sc.validate ? e->err : (u32*)nullptr);
        if (ev) HIPCHK(hipEventRecord(ev[1], st));
    }
    if (e->pend.sample) {                                 // fused-sampling rollouts: actions from / to the side rows
        if (e->step_games == 64 && e->step_wpb == 4) hipLaunchKernelGGL((k_step<64, true, 4>), dim3(blocks(blocks(e->N, 64) + SORT_PAD_WAVES, 4)), dim3(256), 0, st, e->ctx, actions, e->mpk, reward, done, e->err, sc, e->pend, (const u32*)bins);
        else if (e->step_games == 64) hipLaunchKernelGGL((k_step<64, true>), dim3(blocks(e->N, 64) + SORT_PAD_WAVES), dim3(64), 0, st, e->ctx, actions, e->mpk, reward, done, e->err, sc, e->pend, (const u32*)bins);
        else if (e->step_games == 32) hipLaunchKernelGGL((k_step<32, true>), dim3(blocks(e->N, 32) + SORT_PAD_WAVES), dim3(64), 0, st, e->ctx, actions, e->mpk, reward, done, e->err, sc, e->pend, (const u32*)bins);
        else hipLaunchKernelGGL((k_step<16, true>), dim3(blocks(e->N, 16) + SORT_PAD_WAVES), dim3(64), 0, st, e->ctx, actions, e->mpk, reward, done, e->err, sc, e->pend, (const u32*)bins);
        if (ev) HIPCHK(hipEventRecord(ev[2], st));
        HIPCHK(hipGetLastError());
        return CATAN_OK;
    }
    switch (e->step_games) {
    case 16: hipLaunchKernelGGL(k_step<16>, dim3(blocks(e->N, 16) + SORT_PAD_WAVES), dim3(64), 0, st, e->ctx, actions, e->mpk, reward, done, e->err, sc, e->pend, (const u32*)bins); break;
    case 32: hipLaunchKernelGGL(k_step<32>, dim3(blocks(e->N, 32) + SORT_PAD_WAVES), dim3(64), 0, st, e->ctx, actions, e->mpk, reward, done, e->err, sc, e->pend, (const u32*)bins); break;
    default:
        if (e->step_wpb == 4) hipLaunchKernelGGL((k_step<64, false, 4>), dim3(blocks(blocks(e->N, 64) + SORT_PAD_WAVES, 4)), dim3(256), 0, st, e->ctx, actions, e->mpk, reward, done, e->err, sc, e->pend, (const u32*)bins);
        else hipLaunchKernelGGL(k_step<64>, dim3(blocks(e->N, 64) + SORT_PAD_WAVES), dim3(64), 0, st, e->ctx, actions, e->mpk, reward, done, e->err, sc, e->pend, (const u32*)bins);
        break;
    }
    if (ev) HIPCHK(hipEventRecord(ev[2], st));
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
// tier 1 + completion of request list `fl` on stream `st`.  ev (optional): recorded on st: [8] before, [6] after
// workgroups of k_lr_finish (CATAN_LR_GRID: diagnostics, tools/pass_experiments.py).  Deferred schedules (tags >= 2): 3 072 - three tier-1 waves
// per SIMD leave registers for a sampler / k_step wave beside them (4 x 120 of a SIMD's 512 do not), and the launch - one per two passes - has the
// time to take its ~4 000 requests in two rounds: 45.6 -> 44.9 us per pass (profiles/r05_s5_pass_experiments.txt, run 9)
static int lr_grid(bool deferred) {
    static const int g = (getenv("CATAN_LR_GRID") && atoi(getenv("CATAN_LR_GRID")) >= 64) ? atoi(getenv("CATAN_LR_GRID")) : 0;
    return g ? g : (deferred ? LR_GRID_DEFERRED : LR_GRID);
}
static int enqueue_tier1(catan_env_t* e, float* reward, uint8_t* done, hipStream_t st, hipEvent_t* ev, int fl, int lr_budget, bool two_passes = false, int grid = 0) {
    if (grid <= 0 || getenv("CATAN_LR_GRID")) grid = lr_grid(e->pend.ftag >= 2);
    StepCfg sc = step_cfg(e);
    if (ev) HIPCHK(hipEventRecord(ev[8], st));
    if (e->lr_split == 2 || (e->lr_split == 1 && two_passes)) {     // (1: only where a launch has two passes to finish - the library's own deferred loops)
        hipLaunchKernelGGL(k_lr_finish<LRF_SPLIT>, dim3(grid), dim3(64), 0, st, e->ctx, e->mpk, reward, done, sc, e->pend, fl, lr_budget,
                           sc.prof && e->prof_on < 2 ? sc.prof + 2 * PROF_PHASES : nullptr, reinterpret_cast<unsigned long long*>(e->err + 4));
        hipLaunchKernelGGL(k_lr_complete, dim3(LR_COMPLETE_GRID), dim3(64), 0, st, e->ctx, e->mpk, reward, done, sc, e->pend, fl);
    } else
    hipLaunchKernelGGL(k_lr_finish<LRF_TIER1>, dim3(grid), dim3(64), 0, st, e->ctx, e->mpk, reward, done, sc, e->pend, fl, lr_budget,
                       sc.prof && e->prof_on < 2 ? sc.prof + 2 * PROF_PHASES : nullptr, reinterpret_cast<unsigned long long*>(e->err + 4));
    if (ev) HIPCHK(hipEventRecord(ev[6], st));
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
// tier 2 (+ the completion of its games) and the re-deals of window slot pend.sa on stream st.
//   deferred window: the games that ended in k_step / k_lr_finish (list 0) are re-dealt on the side stream while tier 2 runs,
//     those that end in the tier-2 completion (list 1) afterwards.
//   lock-step step: no re-deal is left on the critical path - step_impl started, right behind k_step and on the side stream,
//     the re-deals of the games that ended in k_step AND a speculative successor (shadow records) for every game on the
//     longest-road path; the games that then end in k_lr_finish (list 2) or in the tier-2 completion (list 1) only take
//     their shadow (k_install_list).
// ev (optional): [9] before k_lr_heavy, [3] / [7] after it, [4] at the end
static int enqueue_slow(catan_env_t* e, float* reward, uint8_t* done, hipStream_t st, hipEvent_t* ev, int heavy_grid) {
    StepCfg sc = step_cfg(e);
    const int sa = e->pend.sa; const Limits max_trades = limits_of(e);
    const bool lockstep = heavy_grid == LR_HEAVY_GRID;
    u32* sctr = e->pend.ctr + 8 + 4 * sa;
    u8* busy = e->pend.stag < 2 ? e->pend.busy : nullptr;       // tagged games are released by the sampler, not here
    if (e->cfg.auto_reset) {
        if (!lockstep) {
            HIPCHK(hipEventRecord(e->ev_fork, st));
            HIPCHK(hipStreamWaitEvent(e->side, e->ev_fork, 0));
            hipLaunchKernelGGL(k_reset_list, dim3(RESET_GRID), dim3(64), 0, e->side, e->ctx, e->mpk, max_trades, (const u32*)(sctr + 1),
                               (const i32*)e->pend.resets[sa][0], busy, sc.prof, (const u32*)nullptr, (const u64*)nullptr, (u32*)nullptr, (u32*)nullptr, 0u, e->pend);
            HIPCHK(hipEventRecord(e->ev_join, e->side));
        }
    }
    if (ev) HIPCHK(hipEventRecord(ev[9], st));
    if (!lockstep && e->lr_mid_budget > 0) {     // (both forms of the deferred loop since round 6: the fused loop's failures with it were its own window-close race, deferred_iter)
        // the middle tier: one wave per tier-2 request with a large budget; k_lr_heavy - fewer workgroups: few requests are left - takes the rest
        HIPCHK(hipMemsetAsync(e->pend.ctr + CTR_HEAVY2, 0, sizeof(u32), st));
        hipLaunchKernelGGL(k_lr_finish<LRF_MID>, dim3(LR_MID_GRID), dim3(64), 0, st, e->ctx, e->mpk, reward, done, sc, e->pend, 0, e->lr_mid_budget,
                           (unsigned long long*)nullptr, (unsigned long long*)nullptr);
        hipLaunchKernelGGL(k_lr_heavy, dim3(e->lr_mid_heavy_grid), dim3(LR_HEAVY_THREADS), 0, st, e->ctx, (const u32*)(e->pend.ctr + CTR_HEAVY2), (const u64*)e->pend.heavy2,
                           e->pend.len, e->lr_round[1], e->mpk, reward, done, sc, e->pend);
    } else
    hipLaunchKernelGGL(k_lr_heavy, dim3(heavy_grid), dim3(LR_HEAVY_THREADS), 0, st, e->ctx, (const u32*)sctr, (const u64*)e->pend.heavy[sa], e->pend.len,
                       e->lr_round[lockstep ? 0 : 1], e->mpk, reward, done, sc, e->pend);   // search + completion
    if (ev) HIPCHK(hipEventRecord(ev[3], st));
    if (ev) HIPCHK(hipEventRecord(ev[7], st));
    if (e->cfg.auto_reset) {
        if (lockstep) {
            HIPCHK(hipStreamWaitEvent(st, e->ev_join, 0));      // the side stream's launch of step_impl: it had the whole slow path to finish
            hipLaunchKernelGGL(k_install_list, dim3(256), dim3(64), 0, st, e->ctx, e->mpk, max_trades, (const u32*)(sctr + 3), (const i32*)e->pend.resets[sa][2], busy,
                               (const u32*)e->spec_state, (const u32*)e->spec_mpk, e->spec_epoch, e->err);
            hipLaunchKernelGGL(k_install_list, dim3(256), dim3(64), 0, st, e->ctx, e->mpk, max_trades, (const u32*)(sctr + 2), (const i32*)e->pend.resets[sa][1], busy,
                               (const u32*)e->spec_state, (const u32*)e->spec_mpk, e->spec_epoch, e->err);
        } else {
            HIPCHK(hipStreamWaitEvent(st, e->ev_join, 0));
            hipLaunchKernelGGL(k_reset_list, dim3(RESET_GRID), dim3(64), 0, st, e->ctx, e->mpk, max_trades, (const u32*)(sctr + 2),
                               (const i32*)e->pend.resets[sa][1], busy, sc.prof, (const u32*)nullptr, (const u64*)nullptr, (u32*)nullptr, (u32*)nullptr, 0u, e->pend);
        }
    }
    if (ev) HIPCHK(hipEventRecord(ev[4], st));
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
constexpr int EV_PER_STEP = 10;
// sample_step != nullptr: the random policy draws the actions first (into `actions`), fused with the sort's histogram
static int step_impl(catan_env_t* e, int32_t* actions, float* reward, uint8_t* done, hipStream_t st, hipEvent_t* ev = nullptr,
                     const uint32_t* sample_step = nullptr) {
    e->pend.fa = 0; e->pend.ftag = 1; e->pend.sa = 0; e->pend.stag = 1; e->pend.sample = 0; e->pend.brel = -1; e->pend.nsub = 1;
    { int rr = mt_refill(e, st); if (rr != CATAN_OK) return rr; }
    if (ev) HIPCHK(hipEventRecord(ev[5], st));
    // the step's first kernel zeroes the slow-path list counters (ctr[4..15]) and k_step the count set of the NEXT sort, so a
    // memset is only needed after anything else used the counters (creation, a deferred rollout)
    if (!e->ctr_clean) { HIPCHK(hipMemsetAsync(e->pend.ctr, 0, CTR_WORDS * sizeof(u32), st)); e->ctr_clean = 1; e->lock_parity = 0; }
    e->pend.bsel = e->lock_parity; e->pend.bclear = e->lock_parity ^ 1; e->lock_parity ^= 1;
    if (sample_step)
        hipLaunchKernelGGL(k_sample_random<BLOCK>, dim3(blocks(e->n, BLOCK)), dim3(BLOCK), 0, st, e->ctx, (const u32*)e->mpk, *sample_step, actions,
                           (u32*)nullptr, (u8*)nullptr, 0, 0, e->pend.ctr + 4, 12, e->pend.ctr + 16 + NBINS * e->pend.bsel,
                           e->pend.lists + (size_t)e->pend.bsel * NBINS * e->N);
    int r = enqueue_fast(e, actions, reward, done, st, ev, sample_step != nullptr);
    if (r == CATAN_OK && e->cfg.auto_reset) {                    // the games that ended in k_step: re-dealt on the side stream from here on
        HIPCHK(hipEventRecord(e->ev_fork, st));
        HIPCHK(hipStreamWaitEvent(e->side, e->ev_fork, 0));
        // ... and every game on the longest-road path that this step may end (k_step's list pend.spec) gets a speculative successor
        e->spec_epoch++;
        hipLaunchKernelGGL(k_reset_list, dim3(RESET_GRID), dim3(64), 0, e->side, e->ctx, e->mpk, limits_of(e),
                           (const u32*)(e->pend.ctr + 8 + 1), (const i32*)e->pend.resets[0][0], e->pend.busy, step_cfg(e).prof,
                           (const u32*)(e->pend.ctr + 6), (const u64*)e->pend.spec, e->spec_state, e->spec_mpk, e->spec_epoch, e->pend);
        HIPCHK(hipEventRecord(e->ev_join, e->side));
    }
    if (r == CATAN_OK) r = enqueue_tier1(e, reward, done, st, ev, 0, e->lr_budget[0]);
    if (r == CATAN_OK) r = enqueue_slow(e, reward, done, st, ev, LR_HEAVY_GRID);
    return r;
}

int catan_step(catan_env_t* e, const int32_t* actions, float* reward, uint8_t* done, catan_stream_t stream) {
    if (!e || !actions || !reward || !done) return fail(CATAN_EINVAL, "catan_step: null argument");
    NOT_DEFERRED(e, "catan_step");
    return step_impl(e, const_cast<int32_t*>(actions), reward, done, S(stream));
}

int catan_masks(catan_env_t* e, float* out, catan_stream_t stream) {
    if (!e || !out) return fail(CATAN_EINVAL, "catan_masks: null argument");
    long total = e->n * MASK_BITS;
    if (e->d_it > 0)                     // an open deferred sequence: a waiting game's row is the placeholder (only EndTurn), never a half-written mask
        hipLaunchKernelGGL(k_expand_masks_of, dim3(blocks(total, BLOCK)), dim3(BLOCK), 0, S(stream), (const u32*)e->mpk, (const i32*)nullptr, (long)e->n, (long)e->n, (const u8*)e->pend.busy, out);
    else
        hipLaunchKernelGGL(k_expand_masks, dim3(blocks(total, BLOCK)), dim3(BLOCK), 0, S(stream), e->mpk, e->N, e->n, out);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

int catan_masks_packed_copy(catan_env_t* e, uint32_t* out, catan_stream_t stream) {
    if (!e || !out) return fail(CATAN_EINVAL, "catan_masks_packed_copy: null argument");
    hipLaunchKernelGGL(k_copy_masks11, dim3(blocks(e->n * 11, BLOCK)), dim3(BLOCK), 0, S(stream), (const u32*)e->mpk, (long)e->n, out);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

int catan_masked_row_store(void* dst, const void* src, const int64_t* t, const uint8_t* sel, int64_t rows, int64_t row_bytes,
                           int64_t step_stride_bytes, catan_stream_t stream) {
    if (!dst || !src || !t || !sel || rows <= 0 || row_bytes <= 0 || step_stride_bytes < rows * row_bytes)
        return fail(CATAN_EINVAL, "catan_masked_row_store: bad arguments");
    hipLaunchKernelGGL(k_masked_row_store, dim3((unsigned)rows), dim3(256), 0, S(stream), (unsigned char*)dst, (const unsigned char*)src,
                       (const long long*)t, sel, (long)row_bytes, (long)step_stride_bytes);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

int catan_expand_masks(const uint32_t* packed, int64_t rows, int32_t pitch_words, float* out, catan_stream_t stream) {
    if (!packed || !out || rows <= 0 || pitch_words < (MASK_BITS + 31) / 32) return fail(CATAN_EINVAL, "catan_expand_masks: bad arguments");
    hipLaunchKernelGGL(k_expand_masks, dim3(blocks(rows * MASK_BITS, BLOCK)), dim3(BLOCK), 0, S(stream), packed, (long)rows, (long)rows, out, (int)pitch_words);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

int catan_masks_packed(catan_env_t* e, const uint32_t** out_ptr, int64_t* out_pitch) {
    if (!e || !out_ptr || !out_pitch) return fail(CATAN_EINVAL, "catan_masks_packed: null argument");
    *out_ptr = e->mpk; *out_pitch = MPK_STRIDE;
    return CATAN_OK;
}

int catan_deciding_seat(catan_env_t* e, int32_t* out, catan_stream_t stream) {
    if (!e || !out) return fail(CATAN_EINVAL, "catan_deciding_seat: null argument");
    hipLaunchKernelGGL(k_deciding, dim3(blocks(e->n, BLOCK)), dim3(BLOCK), 0, S(stream), e->ctx, out, 0);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

int catan_players_turn_sim(catan_env_t* e, int32_t* out, catan_stream_t stream) {
    if (!e || !out) return fail(CATAN_EINVAL, "catan_players_turn_sim: null argument");
    hipLaunchKernelGGL(k_deciding, dim3(blocks(e->n, BLOCK)), dim3(BLOCK), 0, S(stream), e->ctx, out, 1);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

int catan_sample_random_actions(catan_env_t* e, uint32_t step_idx, int32_t* actions, catan_stream_t stream) {
    if (!e || !actions) return fail(CATAN_EINVAL, "catan_sample_random_actions: null argument");
    hipLaunchKernelGGL(k_sample_random<BLOCK>, dim3(blocks(e->n, BLOCK)), dim3(BLOCK), 0, S(stream), e->ctx, e->mpk, step_idx, actions,
                       (u32*)nullptr, (u8*)nullptr, 0, 0, (u32*)nullptr, 0, (u32*)nullptr, (i32*)nullptr);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

int catan_state_export(catan_env_t* e, int32_t* blob, const int64_t* env_idx, int64_t cnt, catan_stream_t stream) {
    if (!e || !blob || cnt <= 0 || (!env_idx && cnt > e->n)) return fail(CATAN_EINVAL, "catan_state_export: bad arguments");
    NOT_DEFERRED(e, "catan_state_export");
    hipLaunchKernelGGL(k_export, dim3(blocks(cnt, BLOCK)), dim3(BLOCK), 0, S(stream), e->ctx, (const long*)env_idx, (long)cnt, blob);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

int catan_state_import(catan_env_t* e, const int32_t* blob, const int64_t* env_idx, int64_t cnt, catan_stream_t stream) {
    if (!e || !blob || cnt <= 0 || (!env_idx && cnt > e->n)) return fail(CATAN_EINVAL, "catan_state_import: bad arguments");
    NOT_DEFERRED(e, "catan_state_import");
    hipLaunchKernelGGL(k_import, dim3(blocks(cnt, BLOCK)), dim3(BLOCK), 0, S(stream), e->ctx, (const long*)env_idx, (long)cnt, blob);
    HIPCHK(hipGetLastError());
    return launch_masks(e, S(stream));
}

int catan_set_reward_f64_buffer(catan_env_t* e, double* reward64) {
    if (!e) return fail(CATAN_EINVAL, "catan_set_reward_f64_buffer: null handle");
    e->reward64 = reward64;
    return CATAN_OK;
}

int catan_set_reward_annealing(catan_env_t* e, double f) {
    if (!e) return fail(CATAN_EINVAL, "catan_set_reward_annealing: null handle");
    e->cfg.reward_annealing_factor = f;
    return CATAN_OK;
}

int64_t catan_invalid_action_count(catan_env_t* e, catan_stream_t stream) {
    if (!e) return -1;
    u32 v = 0;
    if (hipMemcpyAsync(&v, e->err, sizeof v, hipMemcpyDeviceToHost, S(stream)) != hipSuccess) return -1;
    if (hipStreamSynchronize(S(stream)) != hipSuccess) return -1;
    return (int64_t)v;
}

int catan_random_rollout(catan_env_t* e, uint32_t step_idx0, int64_t steps, catan_stream_t stream) {
    if (!e || steps < 0) return fail(CATAN_EINVAL, "catan_random_rollout: bad arguments");
    NOT_DEFERRED(e, "catan_random_rollout");
    for (int64_t s = 0; s < steps; s++) {
        const uint32_t step_idx = step_idx0 + (uint32_t)s;
        int r = step_impl(e, e->scratch_actions, e->scratch_reward, e->scratch_done, S(stream), nullptr, &step_idx);
        if (r != CATAN_OK) return r;
    }
    return CATAN_OK;
}

// The deferred iteration in its round 1-3 form (the default: catan_set_deferred_fused): a sampling + sorting kernel in front of
// every k_step, busy tags cleared by it.  Window w = it / window uses slot w & 1 with tag 4 + (w & 1); the sort's bin sets alternate.
// Tier 1 rotates D slots (request list, busy tag 2 / 3 / 6, event pair) by it % D: tier 1 of pass `it` runs on the side stream during
// the passes that follow and its games play again in pass it + D.  D = 2 is the default.  With D = 2 the loop's period is tied to a
// dependency cycle: k_step(it) -> [event hand-over to the side stream, ~11 us] -> k_lr_finish (~40 us: its slowest search) -> [hand-over
// back, ~11 us] -> sampler(it + 2), i.e. 2 P >= 62 us + sampler + k_step (profiles/r05_k_step_pass_experiments.txt).  D = 3
// (CATAN_T1_DEPTH=3) frees the cycle (P >= 36 us) - measured (same file, session 12): 54.3 -> 52.8 us per pass, but the longest-road
// games sit out one more pass (92.6 -> 90.0 % of the games active): 1.118 G env-steps/s either way.  What keeps the pass above the main
// stream's own 45 us is then the event record / wait packets around every pass (~7 us of gaps) and the side work's share of the CUs.
static int t1_depth() {
    static const int D = (getenv("CATAN_T1_DEPTH") && atoi(getenv("CATAN_T1_DEPTH")) == 2) ? 2 : 3;
    return D;
}
static int deferred_iter_legacy(catan_env_t* e, int64_t it, int64_t iters, int window, hipStream_t st, hipEvent_t* ev) {
    const int D = t1_depth();
    const int fa = (int)(it % D);                              // tier-1 slot of this pass
    const int ftag = fa < 2 ? 2 + fa : 6;                      // (4, 5 are the window slots' tags)
    const int ba = (int)(it & 1);                              // bin-count / list set of this pass
    e->ctr_clean = 0;                                          // (the lock-step path re-initialises the counters after this)
    const int64_t w = it / window;
    const int sa = (int)(w & 1);
    const bool opens = it % window == 0, last = it + 1 == iters, closes = (it + 1) % window == 0 || last;
    if (it >= D) HIPCHK(hipStreamWaitEvent(st, e->ev_fdone[fa], 0));        // tier 1 of iteration it-D is complete
    if (it == 0) HIPCHK(hipMemsetAsync(e->pend.ctr, 0, CTR_WORDS * sizeof(u32), st));
    else if (opens) {
        if (w >= 2) HIPCHK(hipStreamWaitEvent(st, e->ev_sdone[sa], 0));     // the slow path of window w-2 is complete
        HIPCHK(hipMemsetAsync(e->pend.ctr + 8 + 4 * sa, 0, 4 * sizeof(u32), st));
    }
    e->pend.fa = fa; e->pend.ftag = ftag; e->pend.sa = sa; e->pend.stag = 4 + sa; e->pend.bsel = ba; e->pend.bclear = ba ^ 1; e->pend.sample = 0; e->pend.brel = -1; e->pend.nsub = 1;
    if (ev) HIPCHK(hipEventRecord(ev[5], st));
    hipLaunchKernelGGL(k_sample_random<BLOCK>, dim3(blocks(e->n, BLOCK)), dim3(BLOCK), 0, st, e->ctx, (const u32*)e->mpk, 0u, e->scratch_actions,
                       e->pctr, e->pend.busy, ftag, (opens && w >= 2) ? 4 + sa : 0, it == 0 ? (u32*)nullptr : e->pend.ctr + (fa < 2 ? 4 + fa : 7), 1,
                       e->pend.ctr + 16 + NBINS * ba, e->pend.lists + (size_t)ba * NBINS * e->N);
    int r = enqueue_fast(e, e->scratch_actions, e->scratch_reward, e->scratch_done, st, ev, true);
    if (r != CATAN_OK) return r;
    // CATAN_DEBUG_EXTRA_EVENTS=k (diagnostics): k more event records per pass - what a queue packet on the main stream costs
    // (profiles/r05_k_step_pass_experiments.txt: +2.9 .. 3.5 us per pass each; the loop has two, this record and the wait above)
    static const int extra = getenv("CATAN_DEBUG_EXTRA_EVENTS") ? atoi(getenv("CATAN_DEBUG_EXTRA_EVENTS")) : 0;
    for (int k = 0; k < extra; k++) HIPCHK(hipEventRecord(e->ev_fork, st));
    HIPCHK(hipEventRecord(e->ev_fready[fa], st));
    HIPCHK(hipStreamWaitEvent(e->fstream[0], e->ev_fready[fa], 0));
    r = enqueue_tier1(e, e->f_reward, e->f_done, e->fstream[0], ev, fa, e->lr_budget[1]);
    if (r != CATAN_OK) return r;
    HIPCHK(hipEventRecord(e->ev_fdone[fa], e->fstream[0]));
    if (closes) {
        // the window's tier-2 / re-deal lists are complete once the outstanding tier-1 launches are (one side stream: the latest implies the others)
        for (int k = 0; k < D && k <= it; k++) HIPCHK(hipStreamWaitEvent(e->sstream, e->ev_fdone[(fa + D - k) % D], 0));
        r = enqueue_slow(e, e->s_reward, e->s_done, e->sstream, ev, LR_HEAVY_GRID_DEFERRED);
        if (r != CATAN_OK) return r;
        HIPCHK(hipEventRecord(e->ev_sdone[sa], e->sstream));
        if (last) {                                   // the call returns with every step complete and no busy game
            HIPCHK(hipStreamWaitEvent(st, e->ev_sdone[sa], 0));
            if (w >= 1) HIPCHK(hipStreamWaitEvent(st, e->ev_sdone[sa ^ 1], 0));
            hipLaunchKernelGGL(k_release_tags, dim3(blocks(e->N, BLOCK)), dim3(BLOCK), 0, st, e->ctx, e->pend.busy);
            e->pend.fa = 0; e->pend.ftag = 1; e->pend.sa = 0; e->pend.stag = 1;
        }
    }
    return r;
}

// One iteration of the deferred rollout (schedule above), fused-sampling form (round 4): the main stream runs ONE kernel per
// pass.  k_step takes each game's action from the game's side row, and for every game it completes it draws the next action
// from the new masks and appends the game to the next pass's lists; a game on the slow path is simply in no list until the
// kernel that completes its step (k_lr_finish: the pass after next; tier 2 / re-deal: the window after next, through the
// window's release list) has drawn its next action and enqueued it.  Every draw is a pure function of (game, decision index,
// state), so the games' trajectories are the lock-step ones whoever draws.  Three sets of bin counts / lists / tier-1 request
// lists rotate (pass % 3): k_step(t) reads set t, appends to set t + 1 and zeroes set t + 2, which k_lr_finish(t) and
// k_step(t + 1) then fill.
// The same iteration with tier 1 forked once per GROUP of two passes (the default since round 5; catan_env::t1_group, CATAN_T1_GROUP=1 for the form above): both passes push their
// longest-road requests to the group's list under the group's tag, tier 1 runs behind the second pass's k_step, and the group's games
// play again when the slot is used next (two groups later: the games of the first pass sit out three passes, those of the second two).
// The event record behind k_step and the event wait in front of the sampler - ~3.3 us of drained main stream each - are then paid
// once per two passes.  A window's last pass (and a call's last) closes its group early, so the window's slow path still sees every
// tier-1 launch that can hand it work.
static int deferred_iter_grouped(catan_env_t* e, int64_t it, int64_t iters, int window, hipStream_t st, hipEvent_t* ev) {
    constexpr int D = 2;                                       // group slots
    const int ba = (int)(it & 1);
    e->ctr_clean = 0;
    const int64_t w = it / window;
    const int sa = (int)(w & 1);
    const bool opens = it % window == 0, last = it + 1 == iters, closes = (it + 1) % window == 0 || last;
    if (it == 0) { e->g_open = 0; e->g_count = 0; }
    const bool g_opens = !e->g_open;
    if (g_opens) { e->g_slot = (int)(e->g_count % D); e->g_passes = 0; e->g_open = 1; }
    const int fa = e->g_slot, ftag = 2 + fa;
    if (g_opens && e->g_count >= D) HIPCHK(hipStreamWaitEvent(st, e->ev_fdone[fa], 0));   // tier 1 of the group that used this slot last is complete
    if (it == 0) HIPCHK(hipMemsetAsync(e->pend.ctr, 0, CTR_WORDS * sizeof(u32), st));
    else if (opens) {
        if (w >= 2) HIPCHK(hipStreamWaitEvent(st, e->ev_sdone[sa], 0));
        HIPCHK(hipMemsetAsync(e->pend.ctr + 8 + 4 * sa, 0, 4 * sizeof(u32), st));
    }
    e->pend.fa = fa; e->pend.ftag = ftag; e->pend.sa = sa; e->pend.stag = 4 + sa; e->pend.bsel = ba; e->pend.bclear = ba ^ 1; e->pend.sample = 0; e->pend.brel = -1; e->pend.nsub = 1;
    if (ev) HIPCHK(hipEventRecord(ev[5], st));
    // the group's first pass releases the slot's previous games (tag) and empties its request list; the second touches neither
    hipLaunchKernelGGL(k_sample_random<512>, dim3(blocks(e->n, 512)), dim3(512), 0, st, e->ctx, (const u32*)e->mpk, 0u, e->scratch_actions,
                       e->pctr, e->pend.busy, g_opens ? ftag : 0, (opens && w >= 2) ? 4 + sa : 0,
                       (g_opens && it != 0) ? e->pend.ctr + 4 + fa : (u32*)nullptr, 1,
                       e->pend.ctr + 16 + NBINS * ba, e->pend.lists + (size_t)ba * NBINS * e->N);
    int r = enqueue_fast(e, e->scratch_actions, e->scratch_reward, e->scratch_done, st, ev, true);
    if (r != CATAN_OK) return r;
    e->g_passes++;
    if (e->g_passes == 2 || closes) {                          // close the group: its tier 1 on the side stream
        HIPCHK(hipEventRecord(e->ev_fready[fa], st));
        HIPCHK(hipStreamWaitEvent(e->fstream[0], e->ev_fready[fa], 0));
        r = enqueue_tier1(e, e->f_reward, e->f_done, e->fstream[0], ev, fa, e->lr_budget[1], true);
        if (r != CATAN_OK) return r;
        HIPCHK(hipEventRecord(e->ev_fdone[fa], e->fstream[0]));
        e->g_open = 0; e->g_count++;
    }
    if (closes) {
        for (int k = 0; k < D && k < e->g_count; k++) HIPCHK(hipStreamWaitEvent(e->sstream, e->ev_fdone[k], 0));
        r = enqueue_slow(e, e->s_reward, e->s_done, e->sstream, ev, LR_HEAVY_GRID_DEFERRED);
        if (r != CATAN_OK) return r;
        HIPCHK(hipEventRecord(e->ev_sdone[sa], e->sstream));
        if (last) {
            HIPCHK(hipStreamWaitEvent(st, e->ev_sdone[sa], 0));
            if (w >= 1) HIPCHK(hipStreamWaitEvent(st, e->ev_sdone[sa ^ 1], 0));
            hipLaunchKernelGGL(k_release_tags, dim3(blocks(e->N, BLOCK)), dim3(BLOCK), 0, st, e->ctx, e->pend.busy);
            e->pend.fa = 0; e->pend.ftag = 1; e->pend.sa = 0; e->pend.stag = 1;
        }
    }
    return r;
}

static int deferred_iter(catan_env_t* e, int64_t it, int64_t iters, int window, hipStream_t st, hipEvent_t* ev) {
    if (!e->deferred_fused && e->t1_group == 2) return deferred_iter_grouped(e, it, iters, window, st, ev);
    if (!e->deferred_fused) return deferred_iter_legacy(e, it, iters, window, st, ev);
    // Tier 1 is forked once per GROUP of P passes (P = 2): a k_lr_finish launch lasts as long as its slowest search (~45 us next
    // to k_step) whatever the number of requests, the launches of consecutive groups serialise on one side stream, and the games
    // of group g return in the first pass of group g + 2 - with P = 1 the chain k_step -> k_lr_finish -> k_step two passes later
    // and the side stream's throughput bound the pass at ~(k_step + k_lr_finish) / 2; with P = 2 neither does.
    // S = P + 2 sets of bin counts / lists rotate: k_step(t) reads set t % S, appends to set (t + 1) % S and empties set
    // (t - 1) % S (its reader is done), which k_lr_finish(g) - launched behind the last pass of group g - and the k_steps before
    // pass (g + 2) P then fill.  Three tier-1 request lists rotate by group.
    static const int P = (getenv("CATAN_T1_PERIOD") && atoi(getenv("CATAN_T1_PERIOD")) == 1) ? 1 : 2;
    const int S = P + 2;
    const int64_t g = it / P;
    const int ga = (int)(g & 1), gl = (int)(g % 3), rs = (int)(it % S);
    const bool gfirst = it % P == 0, glast = (it + 1) % P == 0;
    e->ctr_clean = 0;                                          // (the lock-step path re-initialises the counters after this)
    const int64_t w = it / window;
    const int sa = (int)(w & 1);
    const bool opens = it % window == 0, last = it + 1 == iters, closes = (it + 1) % window == 0 || last;
    if (gfirst && g >= 2) HIPCHK(hipStreamWaitEvent(st, e->ev_fdone[ga], 0));   // tier 1 of group g-2 is complete (its games are in this pass's lists)
    if (ev) HIPCHK(hipEventRecord(ev[5], st));
    e->pend.fa = gl; e->pend.ftag = 2 + ga; e->pend.sa = sa; e->pend.stag = 4 + sa;
    e->pend.bsel = rs; e->pend.bnext = (rs + 1) % S; e->pend.bclear = (rs + S - 1) % S; e->pend.sample = 1; e->pend.nsub = e->fused_subs;
    if (it == 0) {
        HIPCHK(hipMemsetAsync(e->pend.ctr, 0, CTR_WORDS * sizeof(u32), st));
        HIPCHK(hipMemsetAsync(e->pend.bctr, 0, (size_t)BIN_SETS * NBINS * MAX_SUBS * BCTR_PAD * sizeof(u32), st));
        hipLaunchKernelGGL(k_sample_first, dim3(blocks(e->N, BLOCK)), dim3(BLOCK), 0, st, e->ctx, e->mpk, (const u32*)e->pctr, e->pend, rs);
    } else if (opens) {
        if (w >= 2) {                                                       // the slow path of window w-2 is complete: its games play again
            HIPCHK(hipStreamWaitEvent(st, e->ev_sdone[sa], 0));
            hipLaunchKernelGGL(k_release_window, dim3(64), dim3(BLOCK), 0, st, e->ctx, (const u32*)e->mpk, (const u32*)(e->pend.ctr + 11 + 4 * sa),
                               (const i32*)e->pend.resets[sa][2], e->pend, rs);
        }
        HIPCHK(hipMemsetAsync(e->pend.ctr + 8 + 4 * sa, 0, 4 * sizeof(u32), st));
    }
    e->pend.lrq_clear = glast ? (gl + 1) % 3 : -1;             // the next group's request list (read last by tier 1 of group g-2: complete)
    e->pend.brel = (int)(((g + 2) * P) % S);                   // tier 1 of this group: its games return in the first pass of group g + 2
    // CATAN_DEBUG_STEP_DELAY_US=k (diagnostics): the k_step of a pass that closes a window in the middle of its group starts k microseconds late
    static const int dbg_delay_us = getenv("CATAN_DEBUG_STEP_DELAY_US") ? atoi(getenv("CATAN_DEBUG_STEP_DELAY_US")) : 0;
    if (dbg_delay_us > 0 && closes && !(glast || last)) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, (long long)dbg_delay_us * 100);
    int r = enqueue_fast(e, e->scratch_actions, e->scratch_reward, e->scratch_done, st, ev, true);
    if (r != CATAN_OK) return r;
    if (glast || last) {
        static const bool t1_serial = getenv("CATAN_T1_SERIAL") != nullptr;     // (diagnostics: tier 1 on the main stream, no overlap: k_step alone)
        hipStream_t fs = t1_serial ? st : e->fstream[ga];
        HIPCHK(hipEventRecord(e->ev_fready[ga], st));
        HIPCHK(hipStreamWaitEvent(fs, e->ev_fready[ga], 0));
        // (search + lane-per-game completion only with CATAN_LR_SPLIT=2: measured 41.0 us per pass with the split against 38.9 without - here the
        // completion kernel samples and enqueues lane per game with one atomic each, and tier 1's launches are not what bounds this loop)
        // tier 1 staggered behind the next pass's dispatch (T1_STAGGER_US above; CATAN_T1_DELAY_US=k: k microseconds, 0: not staggered)
        static const int t1_delay_us = getenv("CATAN_T1_DELAY_US") ? atoi(getenv("CATAN_T1_DELAY_US")) : T1_STAGGER_US;
        // (the instrumented loop - catan_random_rollout_timed - has an event record in front of every k_step, which holds that launch back by ~3 us: the
        // stagger is lengthened by 4 us there, so that the ORDER of dispatch - and with it k_step's duration by the events - is the uninstrumented
        // loop's: 30.2 us by the events against 30.1 us by rocprofv3 over the plain loop; without the compensation the events read 33 us)
        const int t1_us = t1_delay_us > 0 && ev != nullptr ? t1_delay_us + 4 : t1_delay_us;
        if (t1_us > 0 && !t1_serial) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, fs, (long long)t1_us * 100);
        r = enqueue_tier1(e, e->f_reward, e->f_done, fs, ev, gl, e->lr_budget[2], false, LR_GRID_FUSED);
        if (r != CATAN_OK) return r;
        HIPCHK(hipEventRecord(e->ev_fdone[ga], fs));
    }
    if (closes) {
        // the window's tier-2 / re-deal lists are complete once the outstanding tier-1 launches are ...
        HIPCHK(hipStreamWaitEvent(e->sstream, e->ev_fdone[ga], 0));
        if (g >= 1) HIPCHK(hipStreamWaitEvent(e->sstream, e->ev_fdone[ga ^ 1], 0));
        // ... AND this pass's k_step, which appends the games it ends to the window's re-deal list.  When the pass is its group's last, the
        // tier-1 launch behind it (ev_fdone above) orders the two; a window of an ODD number of passes closes in the MIDDLE of a group, and
        // until round 6 nothing did: the re-deal kernel could read the list's length while k_step was still appending, a game that ended in
        // that pass was then never re-dealt and dropped out of every list (its counter is zeroed when the slot opens again) - the intermittent
        // trajectory-parity failures of the windows of 1 and 5 passes (HISTORY.md round 5; reproduced at will with CATAN_DEBUG_STEP_DELAY_US
        // + CATAN_DEBUG_FUSED_CLOSE_UNORDERED, profiles/r06_fused_close_race.txt).
        static const bool dbg_unordered = getenv("CATAN_DEBUG_FUSED_CLOSE_UNORDERED") != nullptr;    // (diagnostics: the round-4/5 ordering)
        if (!(glast || last) && !dbg_unordered) {
            HIPCHK(hipEventRecord(e->ev_fready[2], st));
            HIPCHK(hipStreamWaitEvent(e->sstream, e->ev_fready[2], 0));
        }
        e->pend.brel = -1;                                     // tier 2 / re-deals: into the window's release list
        r = enqueue_slow(e, e->s_reward, e->s_done, e->sstream, ev, LR_HEAVY_GRID_DEFERRED);
        if (r != CATAN_OK) return r;
        HIPCHK(hipEventRecord(e->ev_sdone[sa], e->sstream));
        if (last) {                                   // the call returns with every step complete and no game left waiting
            HIPCHK(hipStreamWaitEvent(st, e->ev_sdone[sa], 0));
            if (w >= 1) HIPCHK(hipStreamWaitEvent(st, e->ev_sdone[sa ^ 1], 0));
            hipLaunchKernelGGL(k_finish_rollout, dim3(blocks(e->N, BLOCK)), dim3(BLOCK), 0, st, e->ctx, (const u32*)e->mpk, e->pctr, e->pend.busy);
            e->pend.fa = 0; e->pend.ftag = 1; e->pend.sa = 0; e->pend.stag = 1; e->pend.sample = 0; e->pend.nsub = 1;
        }
    }
    return r;
}

int catan_random_rollout_deferred(catan_env_t* e, int64_t iters, int32_t window, catan_stream_t stream) {
    if (!e || iters < 0 || window <= 0) return fail(CATAN_EINVAL, "catan_random_rollout_deferred: bad arguments");
    NOT_DEFERRED(e, "catan_random_rollout_deferred");
    NOT_MT(e, "catan_random_rollout_deferred");
    for (int64_t it = 0; it < iters; it++) {
        int r = deferred_iter(e, it, iters, window, S(stream), nullptr);
        if (r != CATAN_OK) return r;
    }
    return CATAN_OK;
}

// ---- the deferred schedule for caller-supplied actions.  Call `it` of a sequence is iteration `it` of deferred_iter_legacy with
// k_classify_deferred in the sampler's place; what differs is WHEN the waiting games are released: at the end of the call before
// the one they play in again (k_deliver), not at its start - the caller needs their completed state (observation, masks) to choose
// the action it passes to that call.  Every kernel writes a step's reward / done into the handle's result rows (f_reward /
// f_done: one row per game, written by whichever kernel completes the game's step); k_deliver hands them over.
static int deferred_open_error(const catan_env_t* e, const char* what) {
    if (e && e->d_it > 0) return fail(CATAN_EINVAL, std::string(what) + ": a deferred step sequence is open (call catan_step_flush first)");
    return CATAN_OK;
}
int catan_step_deferred(catan_env_t* e, const int32_t* actions, int32_t window, float* reward, uint8_t* done, uint8_t* status, catan_stream_t stream) {
    if (!e || !actions || !reward || !done || !status || window <= 0) return fail(CATAN_EINVAL, "catan_step_deferred: bad arguments");
    NOT_MT(e, "catan_step_deferred");
    hipStream_t st = S(stream);
    if (e->d_it > 0 && (window != e->d_window || st != e->d_stream))
        return fail(CATAN_EINVAL, "catan_step_deferred: window and stream must stay the same between two flushes");
    const int64_t it = e->d_it;
    const int fa = (int)(it & 1);
    const int64_t w = it / window;
    const int sa = (int)(w & 1);
    const bool opens = it % window == 0, closes = (it + 1) % window == 0;
    e->ctr_clean = 0;                                          // (the lock-step path re-initialises the counters after this)
    e->d_window = window; e->d_stream = st;
    // (tier 1 of call it-2 and, when this call opens window w, the slow path of window w-2 were joined at the end of call it-1)
    if (it == 0) HIPCHK(hipMemsetAsync(e->pend.ctr, 0, CTR_WORDS * sizeof(u32), st));
    else if (opens) HIPCHK(hipMemsetAsync(e->pend.ctr + 8 + 4 * sa, 0, 4 * sizeof(u32), st));
    e->pend.fa = fa; e->pend.ftag = 2 + fa; e->pend.sa = sa; e->pend.stag = 4 + sa; e->pend.bsel = fa; e->pend.bclear = fa ^ 1; e->pend.sample = 0; e->pend.brel = -1; e->pend.nsub = 1;
    hipLaunchKernelGGL(k_classify_deferred, dim3(blocks(e->N, BLOCK)), dim3(BLOCK), 0, st, e->ctx, actions, (const u8*)e->pend.busy, e->pend.ctr + 16 + NBINS * fa,
                       e->pend.lists + (size_t)fa * NBINS * e->N, it == 0 ? (u32*)nullptr : e->pend.ctr + 4 + fa, 1, e->cfg.validate_actions ? e->err : (u32*)nullptr);
    int r = enqueue_fast(e, actions, e->f_reward, e->f_done, st, nullptr, true);
    if (r != CATAN_OK) return r;
    HIPCHK(hipEventRecord(e->ev_fready[fa], st));
    HIPCHK(hipStreamWaitEvent(e->fstream[fa], e->ev_fready[fa], 0));
    r = enqueue_tier1(e, e->f_reward, e->f_done, e->fstream[fa], nullptr, fa, e->lr_budget[1]);
    if (r != CATAN_OK) return r;
    HIPCHK(hipEventRecord(e->ev_fdone[fa], e->fstream[fa]));
    if (closes) {                                              // the window's tier-2 / re-deal lists are complete once the outstanding tier-1 launches are
        HIPCHK(hipStreamWaitEvent(e->sstream, e->ev_fdone[fa], 0));
        if (it >= 1) HIPCHK(hipStreamWaitEvent(e->sstream, e->ev_fdone[fa ^ 1], 0));
        r = enqueue_slow(e, e->f_reward, e->f_done, e->sstream, nullptr, LR_HEAVY_GRID_DEFERRED);
        if (r != CATAN_OK) return r;
        HIPCHK(hipEventRecord(e->ev_sdone[sa], e->sstream));
    }
    // what call it+1 needs: tier 1 of call it-1 complete (its games play again), and - if it opens window w' >= 2 - the slow path of w'-2
    const int64_t nit = it + 1, nw = nit / window;
    const bool nopens = nit % window == 0;
    if (nit >= 2) HIPCHK(hipStreamWaitEvent(st, e->ev_fdone[nit & 1], 0));
    if (nopens && nw >= 2) HIPCHK(hipStreamWaitEvent(st, e->ev_sdone[nw & 1], 0));
    hipLaunchKernelGGL(k_deliver, dim3(blocks(e->n, BLOCK)), dim3(BLOCK), 0, st, e->ctx, e->pend.busy, nit >= 2 ? 2 + (int)(nit & 1) : 0, (nopens && nw >= 2) ? 4 + (int)(nw & 1) : 0, 0,
                       (const float*)e->f_reward, (const u8*)e->f_done, reward, done, status);
    HIPCHK(hipGetLastError());
    e->d_it = nit;
    return CATAN_OK;
}

int catan_step_flush(catan_env_t* e, float* reward, uint8_t* done, uint8_t* status, catan_stream_t stream) {
    if (!e || !reward || !done || !status) return fail(CATAN_EINVAL, "catan_step_flush: null argument");
    hipStream_t st = S(stream);
    const int64_t it = e->d_it;
    if (it > 0) {
        if (st != e->d_stream) return fail(CATAN_EINVAL, "catan_step_flush: not the stream of the open sequence");
        const int window = e->d_window;
        const int64_t lw = (it - 1) / window;                  // the window of the last call
        if (it % window != 0) {                                // ... is still open: close it (its pend.sa / stag are still set)
            HIPCHK(hipStreamWaitEvent(e->sstream, e->ev_fdone[0], 0));
            if (it >= 2) HIPCHK(hipStreamWaitEvent(e->sstream, e->ev_fdone[1], 0));
            int r = enqueue_slow(e, e->f_reward, e->f_done, e->sstream, nullptr, LR_HEAVY_GRID_DEFERRED);
            if (r != CATAN_OK) return r;
            HIPCHK(hipEventRecord(e->ev_sdone[lw & 1], e->sstream));
        }
        HIPCHK(hipStreamWaitEvent(st, e->ev_fdone[0], 0));
        if (it >= 2) HIPCHK(hipStreamWaitEvent(st, e->ev_fdone[1], 0));
        HIPCHK(hipStreamWaitEvent(st, e->ev_sdone[lw & 1], 0));
        if (lw >= 1) HIPCHK(hipStreamWaitEvent(st, e->ev_sdone[(lw & 1) ^ 1], 0));
    }
    hipLaunchKernelGGL(k_deliver, dim3(blocks(e->n, BLOCK)), dim3(BLOCK), 0, st, e->ctx, e->pend.busy, 0, 0, 1, (const float*)e->f_reward, (const u8*)e->f_done, reward, done, status);
    HIPCHK(hipGetLastError());
    e->pend.fa = 0; e->pend.ftag = 1; e->pend.sa = 0; e->pend.stag = 1; e->pend.sample = 0; e->pend.nsub = 1;
    e->d_it = 0;
    return CATAN_OK;
}

int catan_policy_counters(catan_env_t* e, uint32_t* out, catan_stream_t stream) {
    if (!e || !out) return fail(CATAN_EINVAL, "catan_policy_counters: null argument");
    HIPCHK(hipMemcpyAsync(out, e->pctr, (size_t)e->n * sizeof(u32), hipMemcpyDeviceToDevice, S(stream)));
    return CATAN_OK;
}

int catan_set_policy_counters(catan_env_t* e, const uint32_t* in, catan_stream_t stream) {
    if (!e) return fail(CATAN_EINVAL, "catan_set_policy_counters: null handle");
    if (in) HIPCHK(hipMemcpyAsync(e->pctr, in, (size_t)e->n * sizeof(u32), hipMemcpyDeviceToDevice, S(stream)));
    else HIPCHK(hipMemsetAsync(e->pctr, 0, (size_t)e->N * sizeof(u32), S(stream)));
    return CATAN_OK;
}

int catan_set_step_wave_games(catan_env_t* e, int32_t games) {
    if (!e || (games != 64 && games != 32 && games != 16)) return fail(CATAN_EINVAL, "catan_set_step_wave_games: 64, 32 or 16");
    e->step_games = games;
    return CATAN_OK;
}

int32_t catan_hip_runtime_version(void) { int v = 0; return hipRuntimeGetVersion(&v) == hipSuccess ? (int32_t)v : -1; }
int32_t catan_step_algorithmic_bytes(void) { return STEP_ALGO_BYTES; }
int32_t catan_step_fused_algorithmic_bytes(void) { return STEP_FUSED_ALGO_BYTES; }
int32_t catan_deferred_fused(const catan_env_t* e) { return e ? e->deferred_fused : -1; }
int catan_set_deferred_fused(catan_env_t* e, int32_t on) {
    if (!e) return fail(CATAN_EINVAL, "catan_set_deferred_fused: null handle");
    e->deferred_fused = on != 0;
    return CATAN_OK;
}
int catan_set_lr_budgets(catan_env_t* e, int32_t lockstep, int32_t deferred) {
    if (!e || lockstep < 1 || deferred < 1) return fail(CATAN_EINVAL, "catan_set_lr_budgets: bad arguments");
    e->lr_budget[0] = lockstep; e->lr_budget[1] = deferred; e->lr_budget[2] = deferred;
    return CATAN_OK;
}

int catan_slow_path_counts(catan_env_t* e, catan_stream_t stream, uint64_t* out3) {
    if (!e || !out3) return fail(CATAN_EINVAL, "catan_slow_path_counts: null argument");
    HIPCHK(hipMemcpyAsync(out3, e->err + 4, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, S(stream)));
    HIPCHK(hipStreamSynchronize(S(stream)));
    return CATAN_OK;
}

int catan_set_lr_rounds(catan_env_t* e, int32_t lockstep, int32_t deferred) {
    if (!e || lockstep < 1 || deferred < 1) return fail(CATAN_EINVAL, "catan_set_lr_rounds: bad arguments");
    e->lr_round[0] = lockstep; e->lr_round[1] = deferred;
    return CATAN_OK;
}

// The rollout loops with a hipEvent around every kernel launch (recorded on the stream the kernel runs on).
// window <= 0: the lock-step loop of catan_random_rollout; window > 0: the deferred loop.  kernel_ms (host, float[5])
// receives the summed elapsed milliseconds of:
// [0] k_sample_random (incl. the sort by action type)  [1] k_step  [2] k_lr_finish  [3] k_lr_heavy (incl. the completion of
// its games)  [4] re-deals / installs (k_reset_list, k_install_list).
int catan_random_rollout_timed(catan_env_t* e, uint32_t step_idx0, int64_t steps, int32_t window, catan_stream_t stream, float* kernel_ms) {
    if (!e || steps <= 0 || !kernel_ms) return fail(CATAN_EINVAL, "catan_random_rollout_timed: bad arguments");
    if (window > 0) NOT_MT(e, "catan_random_rollout_timed (deferred)");
    hipStream_t st = S(stream);
    const int K = EV_PER_STEP;             // see enqueue_fast / enqueue_tier1 / enqueue_slow; [5] = before the sampler
    std::vector<hipEvent_t> ev((size_t)steps * K);
    for (auto& x : ev) HIPCHK(hipEventCreateWithFlags(&x, hipEventDisableSystemFence));   // no L2 flush between kernels
    std::vector<char> slow((size_t)steps, 0);
    for (int64_t s = 0; s < steps; s++) {
        hipEvent_t* v = &ev[(size_t)s * K];
        int r;
        if (window > 0) {
            r = deferred_iter(e, s, steps, window, st, v);
            slow[s] = ((s + 1) % window == 0 || s + 1 == steps);
        } else {
            const uint32_t step_idx = step_idx0 + (uint32_t)s;
            r = step_impl(e, e->scratch_actions, e->scratch_reward, e->scratch_done, st, v, &step_idx);
            slow[s] = 1;
        }
        if (r != CATAN_OK) return r;
    }
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipStreamSynchronize(e->fstream[0]));
    HIPCHK(hipStreamSynchronize(e->sstream));
    for (int k = 0; k < 5; k++) kernel_ms[k] = 0.0f;
    for (int64_t s = 0; s < steps; s++) {
        hipEvent_t* v = &ev[(size_t)s * K];
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, v[5], v[0])); kernel_ms[0] += ms;      // k_sample_random (sampling + sort lists)
        HIPCHK(hipEventElapsedTime(&ms, v[0], v[2])); kernel_ms[1] += ms;      // k_step
        if (hipEventElapsedTime(&ms, v[8], v[6]) == hipSuccess) kernel_ms[2] += ms;   // k_lr_finish (deferred loop: launched once per group of passes)
        else (void)hipGetLastError();
        if (!slow[s]) continue;
        HIPCHK(hipEventElapsedTime(&ms, v[9], v[3])); kernel_ms[3] += ms;      // k_lr_heavy
        HIPCHK(hipEventElapsedTime(&ms, v[7], v[4])); kernel_ms[4] += ms;      // k_reset_list / k_install_list (wait for the side stream + second pass)
    }
    for (auto& x : ev) hipEventDestroy(x);
    return CATAN_OK;
}

int catan_obs(catan_env_t* e, float* out_f, int32_t* out_lists, int32_t* out_lens, catan_stream_t stream) {
    if (!e || !out_f || !out_lists || !out_lens) return fail(CATAN_EINVAL, "catan_obs: null argument");
    static const bool v1 = getenv("CATAN_OBS_V1") != nullptr;          // (the round-1 kernel, kept for A/B timing)
    if (v1) hipLaunchKernelGGL(k_obs, dim3(blocks(e->N, 64)), dim3(64), 0, S(stream), e->ctx, out_f, out_lists, out_lens);
    else hipLaunchKernelGGL((k_obs_rows<ObsF32, OBS_OG>), dim3(blocks(e->n, OBS_OG)), dim3(64), 0, S(stream), e->ctx, out_f, out_lists, out_lens,
                            (float*)nullptr, (signed char*)nullptr, (signed char*)nullptr, (const long long*)nullptr, (const u8*)nullptr, (const i32*)nullptr, (long)e->n);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

static int obs_rows_impl(catan_env_t* e, int32_t bf16, void* dense_f, int32_t* dense_lists, int32_t* dense_lens, void* rows_f, int8_t* rows_lists,
                         int8_t* rows_lens, const int64_t* t_idx, const uint8_t* sel, const int32_t* games, int64_t n_rows, catan_stream_t stream);
int catan_obs_rows(catan_env_t* e, int32_t bf16, void* dense_f, int32_t* dense_lists, int32_t* dense_lens, void* rows_f, int8_t* rows_lists,
                   int8_t* rows_lens, const int64_t* t_idx, const uint8_t* sel, catan_stream_t stream) {
    return obs_rows_impl(e, bf16, dense_f, dense_lists, dense_lens, rows_f, rows_lists, rows_lens, t_idx, sel, nullptr, e ? e->n : 0, stream);
}
int catan_obs_rows_of(catan_env_t* e, int32_t bf16, void* dense_f, int32_t* dense_lists, int32_t* dense_lens, void* rows_f, int8_t* rows_lists,
                      int8_t* rows_lens, const int64_t* t_idx, const uint8_t* sel, const int32_t* games, int64_t n_rows, catan_stream_t stream) {
    if (!e || !games || n_rows <= 0 || n_rows > e->n) return fail(CATAN_EINVAL, "catan_obs_rows_of: bad game list");
    return obs_rows_impl(e, bf16, dense_f, dense_lists, dense_lens, rows_f, rows_lists, rows_lens, t_idx, sel, games, n_rows, stream);
}
int catan_masks_of(catan_env_t* e, float* out, const int32_t* games, int64_t n_rows, catan_stream_t stream) {
    if (!e || !out || !games || n_rows <= 0 || n_rows > e->n) return fail(CATAN_EINVAL, "catan_masks_of: bad arguments");
    hipLaunchKernelGGL(k_expand_masks_of, dim3(blocks(n_rows * MASK_BITS, BLOCK)), dim3(BLOCK), 0, S(stream), (const u32*)e->mpk, games, (long)n_rows, (long)e->n,
                       e->d_it > 0 ? (const u8*)e->pend.busy : (const u8*)nullptr, out);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
static int obs_rows_impl(catan_env_t* e, int32_t bf16, void* dense_f, int32_t* dense_lists, int32_t* dense_lens, void* rows_f, int8_t* rows_lists,
                         int8_t* rows_lens, const int64_t* t_idx, const uint8_t* sel, const int32_t* games, int64_t n_rows, catan_stream_t stream) {
    if (!e || (!dense_f && !rows_f)) return fail(CATAN_EINVAL, "catan_obs_rows: no output");
    if ((dense_lists == nullptr) != (dense_lens == nullptr) || (rows_lists == nullptr) != (rows_lens == nullptr))
        return fail(CATAN_EINVAL, "catan_obs_rows: lists and lens come together");
    if (rows_f && (!t_idx || !sel)) return fail(CATAN_EINVAL, "catan_obs_rows: row stores need t_idx and sel");
    if ((rows_lists && !rows_f) || (dense_lists && !dense_f)) return fail(CATAN_EINVAL, "catan_obs_rows: lists without their observation rows");
    static const int og = getenv("CATAN_OBS_OG") ? atoi(getenv("CATAN_OBS_OG")) : OBS_OG;      // (A/B: games per wave 16 / 8 / 4)
    const dim3 grid(blocks(n_rows, og == 16 ? 16 : og == 4 ? 4 : 8));
#define CATAN_OBS_LAUNCH(OT, CT, G) hipLaunchKernelGGL((k_obs_rows<OT, G>), grid, dim3(64), 0, S(stream), e->ctx, (CT*)dense_f, dense_lists, dense_lens, \
                                                       (CT*)rows_f, (signed char*)rows_lists, (signed char*)rows_lens, (const long long*)t_idx, sel, games, (long)n_rows)
    if (bf16) { if (og == 16) CATAN_OBS_LAUNCH(ObsBF16, unsigned short, 16); else if (og == 4) CATAN_OBS_LAUNCH(ObsBF16, unsigned short, 4); else CATAN_OBS_LAUNCH(ObsBF16, unsigned short, 8); }
    else { if (og == 16) CATAN_OBS_LAUNCH(ObsF32, float, 16); else if (og == 4) CATAN_OBS_LAUNCH(ObsF32, float, 4); else CATAN_OBS_LAUNCH(ObsF32, float, 8); }
#undef CATAN_OBS_LAUNCH
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

int catan_longest_path(catan_env_t* e, const int32_t* players, int32_t* out, catan_stream_t stream) {
    if (!e || !players || !out) return fail(CATAN_EINVAL, "catan_longest_path: null argument");
    hipLaunchKernelGGL(k_longest_path, dim3(blocks(e->N, 64)), dim3(64), 0, S(stream), e->ctx, players, out);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

int64_t catan_gae_workspace_doubles(int64_t N) { return 2 * ((N + GAE_BLOCK - 1) / GAE_BLOCK) + 4; }

int catan_gae(const float* rewards, const float* values, const float* masks, int64_t T, int64_t N, double gamma, double lam,
              float* returns, float* adv_raw, double* workspace, double* stats3, catan_stream_t stream) {
    if (!rewards || !values || !masks || !returns || !adv_raw || !workspace || !stats3 || T <= 0 || N <= 0)
        return fail(CATAN_EINVAL, "catan_gae: bad arguments");
    const int nb = (int)((N + GAE_BLOCK - 1) / GAE_BLOCK);
    // python-scalar semantics of the reference: gamma and gamma*lambda are doubles, cast to fp32 where they meet a tensor
    const float gl = (float)(gamma * lam);
    hipLaunchKernelGGL(k_gae, dim3(nb), dim3(GAE_BLOCK), 0, S(stream), rewards, values, masks, (long)T, (long)N, (float)gamma, gl, returns, adv_raw, workspace);
    hipLaunchKernelGGL(k_adv_stats, dim3(1), dim3(GAE_BLOCK), 0, S(stream), (const double*)workspace, nb, (double)T * (double)N, stats3);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

int catan_adv_normalise(float* adv, int64_t total, const double* stats3, catan_stream_t stream) {
    if (!adv || !stats3 || total <= 0) return fail(CATAN_EINVAL, "catan_adv_normalise: bad arguments");
    long nb = (total + GAE_BLOCK - 1) / GAE_BLOCK;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_adv_normalise, dim3((unsigned)nb), dim3(GAE_BLOCK), 0, S(stream), adv, (long)total, stats3);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

int64_t catan_ppo_loss_workspace_doubles(void) { return 2 * PPO_BLOCKS + 1; }

int catan_ppo_loss(const float* logp, const float* old_logp, const float* adv, const float* values, const float* old_values,
                   const float* returns, int64_t B, float clip, float value_coef, int use_norm, float norm_mean, float norm_std,
                   float* losses2, float* d_logp, float* d_values, double* workspace, catan_stream_t stream) {
    if (!logp || !old_logp || !adv || !values || !old_values || !returns || !losses2 || !d_logp || !d_values || !workspace || B <= 0)
        return fail(CATAN_EINVAL, "catan_ppo_loss: bad arguments");
    PpoArgs a{ clip, value_coef, norm_mean, norm_std, use_norm };
    long nb = (B + PPO_THREADS * 4 - 1) / (PPO_THREADS * 4);
    if (nb > PPO_BLOCKS) nb = PPO_BLOCKS;
    hipLaunchKernelGGL(k_ppo_loss, dim3((unsigned)nb), dim3(PPO_THREADS), 0, S(stream), logp, old_logp, adv, values, old_values, returns, (long)B, a, losses2, d_logp,
                       d_values, workspace);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

int catan_attention_fwd(const void* qkv, const int32_t* lens, void* out, int64_t B, int L, int H, int HD, int is_bf16, catan_stream_t stream) {
    if (!qkv || !out || B <= 0) return fail(CATAN_EINVAL, "catan_attention_fwd: bad arguments");
    return is_bf16 ? attn_dispatch<__hip_bfloat16>(false, qkv, lens, nullptr, out, B, L, H, HD, S(stream))
                   : attn_dispatch<float>(false, qkv, lens, nullptr, out, B, L, H, HD, S(stream));
}
int catan_attention_bwd(const void* qkv, const int32_t* lens, const void* dout, void* dqkv, int64_t B, int L, int H, int HD, int is_bf16, catan_stream_t stream) {
    if (!qkv || !dout || !dqkv || B <= 0) return fail(CATAN_EINVAL, "catan_attention_bwd: bad arguments");
    return is_bf16 ? attn_dispatch<__hip_bfloat16>(true, qkv, lens, dout, dqkv, B, L, H, HD, S(stream))
                   : attn_dispatch<float>(true, qkv, lens, dout, dqkv, B, L, H, HD, S(stream));
}

int catan_layer_norm_fwd(const void* x, const float* w, const float* b, void* y, int64_t rows, int D, float eps, int relu, int is_bf16, catan_stream_t stream) {
    if (!x || !w || !b || !y || rows <= 0) return fail(CATAN_EINVAL, "catan_layer_norm_fwd: bad arguments");
    return is_bf16 ? ln_dispatch<__hip_bfloat16>(false, x, w, b, nullptr, y, nullptr, nullptr, rows, D, eps, relu, S(stream))
                   : ln_dispatch<float>(false, x, w, b, nullptr, y, nullptr, nullptr, rows, D, eps, relu, S(stream));
}
int catan_layer_norm_bwd_res(const void* x, const float* w, const float* b, const void* dy, const void* dres, void* dx, float* dw, float* db,
                             int64_t rows, int D, float eps, int relu, int is_bf16, catan_stream_t stream) {
    if (!x || !w || !b || !dy || !dres || !dx || !dw || !db || rows <= 0) return fail(CATAN_EINVAL, "catan_layer_norm_bwd_res: bad arguments");
    if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dres | (uintptr_t)dx) & 15) return fail(CATAN_EINVAL, "catan_layer_norm_bwd_res: buffers must be 16-byte aligned");
    return is_bf16 ? lnw_res_dispatch<__hip_bfloat16>(x, w, b, dy, dres, dx, dw, db, rows, D, eps, relu, S(stream))
                   : lnw_res_dispatch<float>(x, w, b, dy, dres, dx, dw, db, rows, D, eps, relu, S(stream));
}
int catan_layer_norm_bwd(const void* x, const float* w, const float* b, const void* dy, void* dx, float* dw, float* db, int64_t rows, int D,
                         float eps, int relu, int is_bf16, catan_stream_t stream) {
    if (!x || !w || !b || !dy || !dx || !dw || !db || rows <= 0) return fail(CATAN_EINVAL, "catan_layer_norm_bwd: bad arguments");
    return is_bf16 ? ln_dispatch<__hip_bfloat16>(true, x, w, b, dy, dx, dw, db, rows, D, eps, relu, S(stream))
                   : ln_dispatch<float>(true, x, w, b, dy, dx, dw, db, rows, D, eps, relu, S(stream));
}

// k_step phase profile (100 MHz wall_clock64 ticks): enable/zero, then read [8] sums over waves + [8] maxima.
// phases: stage-in, validate+apply, tier-1 longest road, holder logic (+cut), done/reward, reset, masks, write-back.
static bool wgrad_sliced(int in_features, int out_features) {      // wide inputs: column slices of <= 128 (k_wgrad_tr on strided rows)
    return in_features + 1 > 160 && in_features <= 1024 && (in_features & 7) == 0 && (out_features & 7) == 0 && out_features <= 256;
}
int catan_linear_wgrad_supported(int64_t rows, int in_features, int out_features) {
    if (rows >= 1 && wgrad_sliced(in_features, out_features)) return 1;
    return rows >= 1 && in_features >= 1 && out_features >= 1 && in_features + 1 <= 160 && out_features <= 256;
}

int catan_linear_wgrad(const void* x, const void* dy, float* dw, float* db, int64_t rows, int in_features, int out_features,
                       catan_stream_t stream) {
    if (!x || !dy || !dw || !catan_linear_wgrad_supported(rows, in_features, out_features))
        return fail(CATAN_EINVAL, "catan_linear_wgrad: bad arguments / unsupported widths (in + 1 <= 160 or in a multiple of 8 up to 1024; out <= 256)");
    if (((uintptr_t)x | (uintptr_t)dy) & 15) return fail(CATAN_EINVAL, "catan_linear_wgrad: x and dy must be 16-byte aligned");
    if (wgrad_sliced(in_features, out_features)) {
        const int otw = ((out_features + 15) / 16 + 3) / 4;
        for (int c0 = 0; c0 < in_features; c0 += 128) {
            const int W = in_features - c0 < 128 ? in_features - c0 : 128;
            float* b = c0 == 0 ? db : nullptr;                // the bias gradient comes with the first slice
            int r;
            switch (otw) {
            case 1: r = wgrad_launch_slice<1>(x, dy, dw, b, rows, in_features, c0, W, out_features, S(stream)); break;
            case 2: r = wgrad_launch_slice<2>(x, dy, dw, b, rows, in_features, c0, W, out_features, S(stream)); break;
            case 3: r = wgrad_launch_slice<3>(x, dy, dw, b, rows, in_features, c0, W, out_features, S(stream)); break;
            default: r = wgrad_launch_slice<4>(x, dy, dw, b, rows, in_features, c0, W, out_features, S(stream)); break;
            }
            if (r != CATAN_OK) return r;
        }
        return CATAN_OK;
    }
    const int it = (in_features + 1 + 15) / 16, otw = ((out_features + 15) / 16 + 3) / 4;
    switch (otw) {
    case 1: return wgrad_dispatch_it<1>(it, x, dy, dw, db, rows, in_features, out_features, S(stream));
    case 2: return wgrad_dispatch_it<2>(it, x, dy, dw, db, rows, in_features, out_features, S(stream));
    case 3: return wgrad_dispatch_it<3>(it, x, dy, dw, db, rows, in_features, out_features, S(stream));
    default: return wgrad_dispatch_it<4>(it, x, dy, dw, db, rows, in_features, out_features, S(stream));
    }
}

// catan_linear_wgrad for a list of problems (host array) with as few launches as their tile shapes allow: units of the same
// (output tile, input tile) template share launches of up to WG_MAX_UNITS units; a problem the transposing-read kernel does not take
// (widths that are not multiples of 8) is launched on its own as before.
int catan_linear_wgrad_grouped(const catan_wgrad_problem_t* problems, int32_t n, catan_stream_t stream) {
    if (!problems || n <= 0) return fail(CATAN_EINVAL, "catan_linear_wgrad_grouped: bad arguments");
    struct Unit { WgUnit u; int otw, itc; bool tr; long nb; };
    std::vector<Unit> units;
    for (int p = 0; p < n; p++) {
        const catan_wgrad_problem_t& q = problems[p];
        if (!q.x || !q.dy || !q.dw || !catan_linear_wgrad_supported(q.rows, q.in_features, q.out_features))
            return fail(CATAN_EINVAL, "catan_linear_wgrad_grouped: bad problem / unsupported widths");
        if (((uintptr_t)q.x | (uintptr_t)q.dy) & 15) return fail(CATAN_EINVAL, "catan_linear_wgrad_grouped: x and dy must be 16-byte aligned");
        const int I = q.in_features, O = q.out_features;
        if (q.dw_ld < 0 || q.dw_col0 < 0 || (q.dw_ld != 0 && q.dw_col0 + I > q.dw_ld) || (q.dw_ld == 0 && q.dw_col0 != 0))
            return fail(CATAN_EINVAL, "catan_linear_wgrad_grouped: bad dw window (a column offset needs the leading dimension of dw)");
        const long ldw = q.dw_ld ? q.dw_ld : I;
        const int otw = ((O + 15) / 16 + 3) / 4;
        Unit t; long per;
        t.otw = otw;
        if (wgrad_sliced(I, O)) {
            for (int c0 = 0; c0 < I; c0 += 128) {
                const int W = I - c0 < 128 ? I - c0 : 128;
                wgrad_grid(q.rows, W, O, t.nb, per);
                t.u = WgUnit{ (const unsigned short*)q.x, (const unsigned short*)q.dy, q.dw, c0 == 0 ? q.db : nullptr, (long)q.rows, per, (long)I, ldw, W, O, c0, q.dw_col0 + c0, 0, 0 };
                t.itc = 10; t.tr = true;
                units.push_back(t);
            }
            continue;
        }
        const int it = (I + 1 + 15) / 16;
        wgrad_grid(q.rows, I, O, t.nb, per);
        t.u = WgUnit{ (const unsigned short*)q.x, (const unsigned short*)q.dy, q.dw, q.db, (long)q.rows, per, 0L, ldw, I, O, 0, q.dw_col0, 0, 0 };
        t.itc = it <= 2 ? 2 : (it <= 5 ? 5 : 10);
        t.tr = (I & 7) == 0 && (O & 7) == 0;               // 16 B vectors never straddle a row: the transposing-read kernel
        units.push_back(t);
    }
    std::vector<char> done(units.size(), 0);
    for (size_t a = 0; a < units.size(); a++) {
        if (done[a]) continue;
        WgBatch b; b.n = 0; long blocks = 0;
        for (size_t c = a; c < units.size(); c++) {
            if (done[c] || units[c].otw != units[a].otw || units[c].itc != units[a].itc || units[c].tr != units[a].tr) continue;
            if (b.n == WG_MAX_UNITS) break;
            b.u[b.n] = units[c].u; b.u[b.n].block0 = (int)blocks; blocks += units[c].nb; b.n++; done[c] = 1;
        }
        int r = wgrad_launch_grouped_dispatch(units[a].tr, units[a].otw, units[a].itc, b, blocks, S(stream));
        if (r != CATAN_OK) return r;
    }
    return CATAN_OK;
}

int catan_collector_pre(int64_t n, int32_t T, const int64_t* n_obs, const int64_t* actions, int32_t* a_env, uint8_t* live, catan_stream_t stream) {
    if (n <= 0 || T <= 0 || !n_obs || !actions || !a_env || !live) return fail(CATAN_EINVAL, "catan_collector_pre: bad arguments");
    CollectorArgs a;
    memset(&a, 0, sizeof a);
    a.n = n; a.T = T; a.n_obs = (long long*)n_obs; a.actions = (const long long*)actions; a.a_env = a_env; a.live = live;
    hipLaunchKernelGGL(k_collector_pre, dim3(blocks(n, 256)), dim3(256), 0, S(stream), a);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int catan_collector_post(int64_t n, int32_t T, int64_t* counters4, double* racc, uint8_t* flags4, float* term, int64_t* t_obs, const int64_t* active_pid,
                         const int32_t* deciding, const int32_t* n_deciding, const int64_t* actions, const float* logp, const int32_t* pmasks,
                         const float* reward, const double* reward64, const uint8_t* done, int64_t* st_actions, float* st_logp, int32_t* st_amasks,
                         float* st_rewards, float* st_masks, int64_t* n_complete, const uint8_t* waiting_before, const uint8_t* status, catan_stream_t stream) {
    if (n <= 0 || T <= 0 || !counters4 || !racc || !flags4 || !term || !t_obs || !active_pid || !deciding || !n_deciding || !actions || !logp || !pmasks ||
        !reward || !done || !st_actions || !st_logp || !st_amasks || !st_rewards || !st_masks || !n_complete)
        return fail(CATAN_EINVAL, "catan_collector_post: null argument");
    CollectorArgs a;
    memset(&a, 0, sizeof a);
    a.n = n; a.T = T;
    a.n_obs = (long long*)counters4; a.n_msk = a.n_obs + n; a.n_act = a.n_obs + 2 * n; a.n_rew = a.n_obs + 3 * n;
    a.racc = racc; a.done_since = flags4; a.pending_obs = flags4 + n; a.live = flags4 + 2 * n; a.sel = flags4 + 3 * n;
    a.term = term; a.t_obs = (long long*)t_obs; a.active_pid = (const long long*)active_pid;
    a.deciding = deciding; a.n_deciding = n_deciding; a.actions = (const long long*)actions; a.logp = logp; a.pmasks = pmasks;
    a.reward = reward; a.reward64 = reward64; a.done = done;
    a.st_actions = (long long*)st_actions; a.st_logp = st_logp; a.st_amasks = st_amasks; a.st_rewards = st_rewards; a.st_masks = st_masks;
    a.n_complete = (long long*)n_complete;
    if ((waiting_before == nullptr) != (status == nullptr)) return fail(CATAN_EINVAL, "catan_collector_post: waiting_before and status come together");
    a.waiting_before = waiting_before; a.status = status;
    hipLaunchKernelGGL(k_collector_post, dim3(blocks(n, 256)), dim3(256), 0, S(stream), a);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int catan_gather_rows(const void* src, int64_t src_pitch_bytes, const int64_t* idx, int64_t n, void* dst, int64_t dst_pitch_bytes, int64_t row_bytes,
                      catan_stream_t stream) {
    if (!src || !idx || !dst || n <= 0 || row_bytes <= 0 || (row_bytes & 1) || row_bytes > (1 << 30) || (((uintptr_t)src | (uintptr_t)dst | (uintptr_t)src_pitch_bytes | (uintptr_t)dst_pitch_bytes) & 1))
        return fail(CATAN_EINVAL, "catan_gather_rows: rows are an even number of bytes at even addresses");
    const long nb = n < 16384 ? n : 16384;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)nb), dim3(256), 0, S(stream), (const unsigned char*)src, (long)src_pitch_bytes, (const long long*)idx, (long)n,
                       (unsigned char*)dst, (long)dst_pitch_bytes, (int)row_bytes);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int catan_expand_rows(const void* src, const int64_t* inv, int64_t n, void* out, int64_t row_bytes, catan_stream_t stream) {
    if (!src || !inv || !out || n <= 0 || row_bytes <= 0 || (row_bytes & 15) || (((uintptr_t)src | (uintptr_t)out) & 15))
        return fail(CATAN_EINVAL, "catan_expand_rows: rows are whole 16-byte pieces at 16-byte aligned addresses");
    const int chunks = (int)(row_bytes / 16);
    const long total = n * chunks, nb = (total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536;
    hipLaunchKernelGGL(k_expand_rows16, dim3((unsigned)nb), dim3(256), 0, S(stream), (const uint4*)src, (const long long*)inv, (long)n, (uint4*)out, chunks);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int catan_segment_sum_rows(const void* dy, int64_t dy_pitch_bytes, const int64_t* order, const int64_t* start, int64_t segments, void* out, int64_t row_bytes,
                           catan_stream_t stream) {
    if (!dy || !order || !start || !out || segments <= 0 || row_bytes <= 0 || (row_bytes & 15) || (((uintptr_t)dy | (uintptr_t)out | (uintptr_t)dy_pitch_bytes) & 15) ||
        dy_pitch_bytes < row_bytes)
        return fail(CATAN_EINVAL, "catan_segment_sum_rows: rows are whole 16-byte pieces at 16-byte aligned addresses");
    const int chunks = (int)(row_bytes / 16);
    const long total = segments * chunks, nb = (total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536;
    hipLaunchKernelGGL(k_segment_sum16, dim3((unsigned)nb), dim3(256), 0, S(stream), (const uint4*)dy, (const long long*)order, (const long long*)start, (long)segments,
                       (uint4*)out, chunks, (long)(dy_pitch_bytes / 16));
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int catan_concat_rows(const void* const* srcs, const int64_t* row_bytes, int n, void* out, int64_t out_pitch_bytes, int64_t rows, catan_stream_t stream) {
    if (!srcs || !row_bytes || !out || n < 1 || n > CC_MAX || rows <= 0 || (((uintptr_t)out | (uintptr_t)out_pitch_bytes) & 15))
        return fail(CATAN_EINVAL, "catan_concat_rows: 1..4 sources; rows are whole 16-byte pieces at 16-byte aligned addresses");
    ConcatSrc cs;
    cs.n = n;
    int total = 0;
    for (int k = 0; k < CC_MAX; k++) { cs.src[k] = nullptr; cs.chunks[k] = 0; cs.first[k] = 0; }
    for (int k = 0; k < n; k++) {
        if (!srcs[k] || row_bytes[k] <= 0 || (row_bytes[k] & 15) || ((uintptr_t)srcs[k] & 15))
            return fail(CATAN_EINVAL, "catan_concat_rows: 1..4 sources; rows are whole 16-byte pieces at 16-byte aligned addresses");
        cs.src[k] = (const uint4*)srcs[k]; cs.chunks[k] = (int)(row_bytes[k] / 16); cs.first[k] = total; total += cs.chunks[k];
    }
    if ((long)total * 16 > out_pitch_bytes) return fail(CATAN_EINVAL, "catan_concat_rows: the sources' rows exceed the output pitch");
    const long pieces = rows * total, nb = (pieces + 255) / 256 < 65536 ? (pieces + 255) / 256 : 65536;
    hipLaunchKernelGGL(k_concat_rows16, dim3((unsigned)nb), dim3(256), 0, S(stream), cs, (uint4*)out, (long)rows, total, (int)(out_pitch_bytes / 16));
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int catan_recurrent_given(const int64_t* acts, int64_t acts_ld, const float* cur_res, const float* fixed, int32_t kf, int32_t from_hand, int64_t rows, int32_t cond_bf16,
                          void* cond, float* mask, int64_t* given, float* keep, float* out_final, catan_stream_t stream) {
    if (!acts || !cur_res || !cond || !mask || !given || !keep || !out_final || rows <= 0 || acts_ld < 4 || kf < 0 || kf > 32 || (kf > 0 && !fixed))
        return fail(CATAN_EINVAL, "catan_recurrent_given: bad arguments");
    const unsigned nb = (unsigned)((rows + 255) / 256);
    if (cond_bf16) hipLaunchKernelGGL(k_recurrent_given<true>, dim3(nb), dim3(256), 0, S(stream), (const long long*)acts, (long)acts_ld, cur_res, fixed, (int)kf, (int)from_hand, (long)rows,
                                      cond, mask, (long long*)given, keep, out_final);
    else hipLaunchKernelGGL(k_recurrent_given<false>, dim3(nb), dim3(256), 0, S(stream), (const long long*)acts, (long)acts_ld, cur_res, fixed, (int)kf, (int)from_hand, (long)rows,
                            cond, mask, (long long*)given, keep, out_final);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int catan_scatter_rows_ranges(const void* dy, int64_t dy_pitch_bytes, const int64_t* perm, int64_t n_perm, const int64_t* ranges, int n_ranges,
                              const void* add0, const void* add1, void* out, int64_t row_bytes, catan_stream_t stream) {
    if (!dy || !perm || !ranges || !out || n_perm <= 0 || n_ranges < 0 || n_ranges > SR_MAX || row_bytes <= 0 || (row_bytes & 15) ||
        (((uintptr_t)dy | (uintptr_t)out | (uintptr_t)dy_pitch_bytes | (uintptr_t)add0 | (uintptr_t)add1) & 15) || dy_pitch_bytes < row_bytes)
        return fail(CATAN_EINVAL, "catan_scatter_rows_ranges: at most 16 ranges; rows are whole 16-byte pieces at 16-byte aligned addresses");
    ScatterRanges rg;
    rg.n = n_ranges;
    for (int k = 0; k < SR_MAX; k++) { rg.a[k] = 0; rg.b[k] = 0; rg.off[k] = 0; }
    for (int k = 0; k < n_ranges; k++) {
        rg.a[k] = ranges[3 * k]; rg.b[k] = ranges[3 * k + 1]; rg.off[k] = ranges[3 * k + 2];
        if (rg.a[k] < 0 || rg.b[k] < rg.a[k] || rg.b[k] > n_perm || rg.off[k] < 0) return fail(CATAN_EINVAL, "catan_scatter_rows_ranges: a range outside the permutation");
        // the ranges' rows of dy follow each other (nn_kernels.gather_ranges lays them out that way): range k starts where range k - 1 ended, so
        // dy holds exactly the sum of the range lengths rows and no offset can point past it
        if (rg.off[k] != (k ? rg.off[k - 1] + (rg.b[k - 1] - rg.a[k - 1]) : 0)) return fail(CATAN_EINVAL, "catan_scatter_rows_ranges: the ranges' rows of dy must be consecutive from row 0");
    }
    const int chunks = (int)(row_bytes / 16);
    const long total = n_perm * chunks, nb = (total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536;
    hipLaunchKernelGGL(k_scatter_ranges16, dim3((unsigned)nb), dim3(256), 0, S(stream), (const uint4*)dy, (long)(dy_pitch_bytes / 16), (const long long*)perm, (long)n_perm, rg,
                       (const uint4*)add0, (const uint4*)add1, (uint4*)out, chunks);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int catan_ffn_bwd_dx(const void* dx, const void* h, const void* x, const void* w2t, const void* w1t, const float* ln_w, float eps, void* dh, void* dx_out,
                     float* dln_w, float* dln_b, int64_t rows, catan_stream_t stream) {
    if (!dx || !h || !x || !w2t || !w1t || !ln_w || !dh || !dx_out || !dln_w || !dln_b || rows <= 0 ||
        (((uintptr_t)dx | (uintptr_t)h | (uintptr_t)x | (uintptr_t)w2t | (uintptr_t)w1t | (uintptr_t)dh | (uintptr_t)dx_out) & 15))
        return fail(CATAN_EINVAL, "catan_ffn_bwd_dx: null or misaligned argument");
    long nb = ((rows + 15) / 16 + 3) / 4;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_ffn_bwd_dx, dim3((unsigned)nb), dim3(256), 0, S(stream), (const unsigned short*)dx, (const unsigned short*)h, (const unsigned short*)x,
                       (const unsigned short*)w2t, (const unsigned short*)w1t, ln_w, eps, (unsigned short*)dh, (unsigned short*)dx_out, dln_w, dln_b, (long)rows);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int catan_qkv_bwd_dx(const void* dqkv, const void* x, const void* dres, const void* wt, const float* ln_w, float eps, void* dx_out, float* dln_w, float* dln_b,
                     int64_t rows, catan_stream_t stream) {
    if (!dqkv || !x || !dres || !wt || !ln_w || !dx_out || !dln_w || !dln_b || rows <= 0 ||
        (((uintptr_t)dqkv | (uintptr_t)x | (uintptr_t)dres | (uintptr_t)wt | (uintptr_t)dx_out) & 15))
        return fail(CATAN_EINVAL, "catan_qkv_bwd_dx: null or misaligned argument");
    long nb = ((rows + 15) / 16 + 3) / 4;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_qkv_bwd_dx, dim3((unsigned)nb), dim3(256), 0, S(stream), (const unsigned short*)dqkv, (const unsigned short*)x, (const unsigned short*)dres,
                       (const unsigned short*)wt, ln_w, eps, (unsigned short*)dx_out, dln_w, dln_b, (long)rows);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int64_t catan_wgrad_big_workspace_floats(int64_t rows, int in_features, int out_features) {
    if (rows <= 0 || in_features <= 0 || out_features <= 0) return 0;
    const int tiles_i = (in_features + 1 + WB_T - 1) / WB_T;
    const int groups = 8 * (rows >= 8 * 2 * 4096 ? 2 : 1);
    return (int64_t)groups * out_features * tiles_i * WB_T;
}
int catan_linear_wgrad_big(const void* x, const void* dy, float* dw, int64_t dw_ld, float* db, float* workspace, int64_t rows, int in_features,
                           int out_features, int accumulate, catan_stream_t stream) {
    if (!x || !dy || !dw || !workspace || rows <= 0 || in_features < 8 || (in_features & 7) || out_features < WB_T || (out_features % WB_T) || dw_ld < in_features ||
        (((uintptr_t)x | (uintptr_t)dy) & 15))
        return fail(CATAN_EINVAL, "catan_linear_wgrad_big: bad arguments (in a multiple of 8, out a multiple of 128, 16-byte aligned operands)");
    const int tiles_o = out_features / WB_T, tiles_i = (in_features + 1 + WB_T - 1) / WB_T;
    const int groups = 8 * (rows >= 8 * 2 * 4096 ? 2 : 1);                 // row groups: one per XCD, two when every group still has >= 4 096 rows
    long per = (rows + groups - 1) / groups;
    per = (per + WG_KT - 1) / WG_KT * WG_KT;
    const long nblocks = (long)groups * tiles_o * tiles_i;
    hipLaunchKernelGGL(k_wgrad_big, dim3((unsigned)nblocks), dim3(256), 0, S(stream), (const unsigned short*)x, (const unsigned short*)dy, workspace, (long)rows,
                       in_features, out_features, per, tiles_o, tiles_i);
    const long n = (long)out_features * (in_features + 1);
    hipLaunchKernelGGL(k_wgrad_big_reduce, dim3(blocks(n, 256)), dim3(256), 0, S(stream), (const float*)workspace, groups, out_features, in_features, tiles_i * WB_T,
                       dw, (long)dw_ld, db, accumulate);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int32_t catan_adam_chunk_elements(void) { return OPT_CHUNK; }
int catan_adam_step(const void* tensors, const void* chunks, int32_t n_chunks, const void* grads, void* partial, float max_norm, float lr, float beta1,
                    float beta2, float eps, float bias_correction1, float bias_correction2_sqrt, float* norm_out, catan_stream_t stream) {
    static_assert(sizeof(AdamTensor) == sizeof(catan_adam_tensor_t) && sizeof(AdamChunk) == sizeof(catan_adam_chunk_t), "the header's structs are the kernels'");
    if (!tensors || !chunks || !grads || !partial || n_chunks <= 0 || !(bias_correction1 > 0.f) || !(bias_correction2_sqrt > 0.f))
        return fail(CATAN_EINVAL, "catan_adam_step: bad arguments");
    AdamHyper h = { max_norm, lr, beta1, beta2, eps, bias_correction1, bias_correction2_sqrt, max_norm > 0.f ? 1 : 0 };
    hipLaunchKernelGGL(k_grad_sumsq, dim3((unsigned)n_chunks), dim3(OPT_BLOCK), 0, S(stream), (const AdamChunk*)chunks, (const float* const*)grads, (double*)partial);
    hipLaunchKernelGGL(k_adam_step, dim3((unsigned)n_chunks), dim3(OPT_BLOCK), 0, S(stream), (const AdamTensor*)tensors, (const AdamChunk*)chunks, (int)n_chunks,
                       (const float* const*)grads, (const double*)partial, h, norm_out);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int32_t catan_weight_image_bytes(void) { return (int32_t)sizeof(WeightImage); }
int catan_weight_images(const void* table, int32_t n, catan_stream_t stream) {
    if (!table || n <= 0) return fail(CATAN_EINVAL, "catan_weight_images: bad arguments");
    static_assert(sizeof(WeightImage) == 64 && sizeof(catan_weight_image_t) == sizeof(WeightImage), "the header's struct is the kernel's");
    hipLaunchKernelGGL(k_weight_images, dim3((unsigned)n, 8), dim3(256), 0, S(stream), (const WeightImage*)table, (int)n);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int catan_ffn_bwd(const void* dx, const void* h, const void* x, const void* n, const void* w2t, const void* w1t, const float* ln_w, const float* ln_b, float eps,
                  void* dx_out, float* dw2, float* db2, float* dw1, float* db1, float* dln_w, float* dln_b, int64_t rows, catan_stream_t stream) {
    if (!dx || !h || !x || (!n && !ln_b) || !w2t || !w1t || !ln_w || !dx_out || !dw2 || !db2 || !dw1 || !db1 || !dln_w || !dln_b || rows <= 0 ||
        (((uintptr_t)dx | (uintptr_t)h | (uintptr_t)x | (uintptr_t)n | (uintptr_t)w2t | (uintptr_t)w1t | (uintptr_t)dx_out) & 15))
        return fail(CATAN_EINVAL, "catan_ffn_bwd: null or misaligned argument");
    // every block ends with 16 640 atomics: several stages of 64 rows per block, at most 512 blocks (the grid rule of wgrad_grid)
    const long stages = (rows + FW_ROWS - 1) / FW_ROWS;
    long nb = stages / 16 < 1 ? 1 : (stages / 16 < 512 ? stages / 16 : 512);
    const long per = (stages + nb - 1) / nb * FW_ROWS;
    nb = (rows + per - 1) / per;
    FfnOutProj op = { nullptr, nullptr, nullptr, nullptr, nullptr };
    if (n) hipLaunchKernelGGL((k_ffn_bwd_w<false, false>), dim3((unsigned)nb), dim3(256), 0, S(stream), (const unsigned short*)dx, (const unsigned short*)h, (const unsigned short*)x,
                       (const unsigned short*)n, (const unsigned short*)w2t, (const unsigned short*)w1t, ln_w, ln_b, eps, (unsigned short*)dx_out, dw2, db2, dw1, db1,
                       dln_w, dln_b, (long)rows, per, op);
    else hipLaunchKernelGGL((k_ffn_bwd_w<false, true>), dim3((unsigned)nb), dim3(256), 0, S(stream), (const unsigned short*)dx, (const unsigned short*)h, (const unsigned short*)x,
                       (const unsigned short*)n, (const unsigned short*)w2t, (const unsigned short*)w1t, ln_w, ln_b, eps, (unsigned short*)dx_out, dw2, db2, dw1, db1,
                       dln_w, dln_b, (long)rows, per, op);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int catan_ffn_outproj_bwd(const void* dx, const void* h, const void* x, const void* n, const void* w2t, const void* w1t, const float* ln_w, const float* ln_b, float eps,
                          void* dx_out, float* dw2, float* db2, float* dw1, float* db1, float* dln_w, float* dln_b,
                          const void* o, const void* wot, void* d_o, float* dwo, float* dbo, int64_t rows, catan_stream_t stream) {
    if (!dx || !h || !x || (!n && !ln_b) || !w2t || !w1t || !ln_w || !dx_out || !dw2 || !db2 || !dw1 || !db1 || !dln_w || !dln_b || !o || !wot || !d_o || !dwo || !dbo || rows <= 0 ||
        (((uintptr_t)dx | (uintptr_t)h | (uintptr_t)x | (uintptr_t)n | (uintptr_t)w2t | (uintptr_t)w1t | (uintptr_t)dx_out | (uintptr_t)o | (uintptr_t)wot | (uintptr_t)d_o) & 15))
        return fail(CATAN_EINVAL, "catan_ffn_outproj_bwd: null or misaligned argument");
    const long stages = (rows + FW_ROWS - 1) / FW_ROWS;
    long nb = stages / 16 < 1 ? 1 : (stages / 16 < 512 ? stages / 16 : 512);
    const long per = (stages + nb - 1) / nb * FW_ROWS;
    nb = (rows + per - 1) / per;
    FfnOutProj op = { (const unsigned short*)o, (const unsigned short*)wot, (unsigned short*)d_o, dwo, dbo };
    if (n) hipLaunchKernelGGL((k_ffn_bwd_w<true, false>), dim3((unsigned)nb), dim3(256), 0, S(stream), (const unsigned short*)dx, (const unsigned short*)h, (const unsigned short*)x,
                       (const unsigned short*)n, (const unsigned short*)w2t, (const unsigned short*)w1t, ln_w, ln_b, eps, (unsigned short*)dx_out, dw2, db2, dw1, db1,
                       dln_w, dln_b, (long)rows, per, op);
    else hipLaunchKernelGGL((k_ffn_bwd_w<true, true>), dim3((unsigned)nb), dim3(256), 0, S(stream), (const unsigned short*)dx, (const unsigned short*)h, (const unsigned short*)x,
                       (const unsigned short*)n, (const unsigned short*)w2t, (const unsigned short*)w1t, ln_w, ln_b, eps, (unsigned short*)dx_out, dw2, db2, dw1, db1,
                       dln_w, dln_b, (long)rows, per, op);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int catan_ffn_outproj_bwd_rh(const void* dx, const void* x, const void* w2t, const void* w1t, const void* w1, const float* b1, const float* ln_w, const float* ln_b, float eps,
                             void* dx_out, float* dw2, float* db2, float* dw1, float* db1, float* dln_w, float* dln_b,
                             const void* o, const void* wot, void* d_o, float* dwo, float* dbo, int64_t rows, catan_stream_t stream) {
    if (!dx || !x || !w2t || !w1t || !w1 || !b1 || !ln_w || !ln_b || !dx_out || !dw2 || !db2 || !dw1 || !db1 || !dln_w || !dln_b || !o || !wot || !d_o || !dwo || !dbo || rows <= 0 ||
        (((uintptr_t)dx | (uintptr_t)x | (uintptr_t)w2t | (uintptr_t)w1t | (uintptr_t)w1 | (uintptr_t)dx_out | (uintptr_t)o | (uintptr_t)wot | (uintptr_t)d_o) & 15))
        return fail(CATAN_EINVAL, "catan_ffn_outproj_bwd_rh: null or misaligned argument");
    const long stages = (rows + FW_ROWS - 1) / FW_ROWS;
    long nb = stages / 16 < 1 ? 1 : (stages / 16 < 512 ? stages / 16 : 512);
    const long per = (stages + nb - 1) / nb * FW_ROWS;
    nb = (rows + per - 1) / per;
    FfnOutProj op = { (const unsigned short*)o, (const unsigned short*)wot, (unsigned short*)d_o, dwo, dbo };
    FfnRecomputeH rh = { (const unsigned short*)w1, b1 };
    hipLaunchKernelGGL((k_ffn_bwd_w<true, true, true>), dim3((unsigned)nb), dim3(256), 0, S(stream), (const unsigned short*)dx, (const unsigned short*)nullptr, (const unsigned short*)x,
                       (const unsigned short*)nullptr, (const unsigned short*)w2t, (const unsigned short*)w1t, ln_w, ln_b, eps, (unsigned short*)dx_out, dw2, db2, dw1, db1,
                       dln_w, dln_b, (long)rows, per, op, rh);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int catan_qkv_bwd(const void* dqkv, const void* x, const void* dres, const void* n, const void* wt, const float* ln_w, const float* ln_b, float eps, void* dx_out,
                  float* dw, float* db, float* dln_w, float* dln_b, int64_t rows, catan_stream_t stream) {
    if (!dqkv || !x || !dres || (!n && !ln_b) || !wt || !ln_w || !dx_out || !dw || !db || !dln_w || !dln_b || rows <= 0 ||
        (((uintptr_t)dqkv | (uintptr_t)x | (uintptr_t)dres | (uintptr_t)n | (uintptr_t)wt | (uintptr_t)dx_out) & 15))
        return fail(CATAN_EINVAL, "catan_qkv_bwd: null or misaligned argument");
    const long stages = (rows + FW_ROWS - 1) / FW_ROWS;
    long nb = stages / 16 < 1 ? 1 : (stages / 16 < 512 ? stages / 16 : 512);
    const long per = (stages + nb - 1) / nb * FW_ROWS;
    nb = (rows + per - 1) / per;
    if (n) hipLaunchKernelGGL(k_qkv_bwd_w<false>, dim3((unsigned)nb), dim3(256), 0, S(stream), (const unsigned short*)dqkv, (const unsigned short*)x, (const unsigned short*)dres,
                       (const unsigned short*)n, (const unsigned short*)wt, ln_w, ln_b, eps, (unsigned short*)dx_out, dw, db, dln_w, dln_b, (long)rows, per);
    else hipLaunchKernelGGL(k_qkv_bwd_w<true>, dim3((unsigned)nb), dim3(256), 0, S(stream), (const unsigned short*)dqkv, (const unsigned short*)x, (const unsigned short*)dres,
                       (const unsigned short*)n, (const unsigned short*)wt, ln_w, ln_b, eps, (unsigned short*)dx_out, dw, db, dln_w, dln_b, (long)rows, per);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int32_t catan_head_weight_elems(void) { return HD_WELEMS; }
int32_t catan_head_vec_elems(void) { return HD_VELEMS; }
int catan_head_fwd(const void* pre, int64_t pre_ld, const float* cond, int64_t cond_ld, int32_t ncond, const void* wts, const float* vec, float eps,
                   int32_t K, const float* mask, int64_t mask_ld, const float* u, int64_t* action, float* logp, int64_t B, catan_stream_t stream) {
    if (!pre || !wts || !vec || !mask || !action || !logp || B <= 0 || K < 1 || K > HD_KP || ncond < 0 || ncond > HD_NCP || (ncond > 0 && !cond) ||
        pre_ld % 8 != 0 || ((uintptr_t)pre & 15) != 0)
        return fail(CATAN_EINVAL, "catan_head_fwd: bad arguments (K <= 80, ncond <= 32, pre 16-byte aligned with a row pitch that is a multiple of 8)");
    HeadArgs a;
    a.pre = (const unsigned short*)pre; a.pre_ld = pre_ld; a.cond = cond; a.cond_ld = cond_ld; a.ncond = ncond;
    a.wts = (const unsigned short*)wts; a.vec = vec; a.eps = eps; a.K = K; a.mask = mask; a.mask_ld = mask_ld; a.u = u;
    a.action = (long long*)action; a.logp = logp; a.B = B;
    a.state = nullptr; a.head_id = 0; a.step = 0; a.maskmat = nullptr; a.cur_res = nullptr; a.trade = nullptr; a.custom = nullptr;
    a.forced = nullptr; a.actions = nullptr; a.logp_out = nullptr;
    return head_launch(a, S(stream));
}

int32_t catan_head_state_floats(void) { return HD_STATE; }
int catan_head_chain(const void* pre, int64_t pre_ld, const void* wts, const float* vec, float eps, int32_t head_id, int32_t step, float* state,
                     const float* maskmat, const float* cur_res, const float* trade, const float* custom, const int64_t* forced, const float* u,
                     int64_t* actions, float* logp_out, int64_t B, catan_stream_t stream) {
    static const int KS[12] = { 13, 54, 73, 19, 5, 2, 3, 6, 6, 5, 5, 5 }, NC[12] = { 0, 2, 0, 0, 0, 32, 2, 6, 12, 4, 9, 0 };
    if (!pre || !wts || !vec || !state || !maskmat || !cur_res || !actions || !logp_out || B <= 0 || head_id < 0 || head_id > 11 || step < 0 || step > 3 ||
        ((head_id != 7 && head_id != 8) && step != 0) || (head_id == 5 && (!trade || !custom)) || pre_ld % 8 != 0 || ((uintptr_t)pre & 15) != 0 ||
        ((uintptr_t)state & 15) != 0)
        return fail(CATAN_EINVAL, "catan_head_chain: bad arguments");
    HeadArgs a;
    a.pre = (const unsigned short*)pre; a.pre_ld = pre_ld; a.cond = nullptr; a.cond_ld = 0; a.ncond = NC[head_id];
    a.wts = (const unsigned short*)wts; a.vec = vec; a.eps = eps; a.K = KS[head_id]; a.mask = nullptr; a.mask_ld = 0; a.u = u;
    a.action = nullptr; a.logp = nullptr; a.B = B;
    a.state = state; a.head_id = head_id; a.step = step; a.maskmat = maskmat; a.cur_res = cur_res; a.trade = trade; a.custom = custom;
    a.forced = (const long long*)forced; a.actions = (long long*)actions; a.logp_out = logp_out;
    return head_launch(a, S(stream));
}

int catan_randomise_uncertainty(catan_env_t* e, const int32_t* controlling_player, catan_stream_t stream) {
    if (!e || !controlling_player) return fail(CATAN_EINVAL, "catan_randomise_uncertainty: null argument");
    NOT_DEFERRED(e, "catan_randomise_uncertainty");
    NOT_MT(e, "catan_randomise_uncertainty");
    hipLaunchKernelGGL(k_randomise_uncertainty, dim3(blocks(e->n, 64)), dim3(64), 0, S(stream), e->ctx, controlling_player, e->mpk, e->err,
                       100000, limits_of(e));
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

int64_t catan_inconsistent_deal_count(catan_env_t* e, catan_stream_t stream) {
    if (!e) return -1;
    u32 v = 0;
    if (hipMemcpyAsync(&v, e->err + 1, sizeof v, hipMemcpyDeviceToHost, S(stream)) != hipSuccess) return -1;
    if (hipStreamSynchronize(S(stream)) != hipSuccess) return -1;
    return (int64_t)v;
}

int catan_linear_rows_supported(int64_t rows, int in_features, int out_features) {
    if (!(rows >= 1 && in_features >= 8 && in_features <= 192 && (in_features & 7) == 0 && out_features >= 1 && out_features <= 192)) return 0;
    const int ks = (in_features + 31) / 32, nt = (out_features + 15) / 16;
    const int ks_i = ks <= 4 ? ks : 6, nt_i = nt <= 1 ? 1 : (nt <= 2 ? 2 : (nt <= 4 ? 4 : (nt <= 8 ? 8 : 12)));   // instantiated sizes
    return ks_i * nt_i <= 32;                                // W fragments (4 VGPRs each) must fit the register file
}

int catan_linear_rows_fused(const void* x, const void* w, const void* bias, void* y, int64_t rows, int in_features, int out_features,
                            const void* aux, int mode, catan_stream_t stream) {
    if (!x || !w || !y || !catan_linear_rows_supported(rows, in_features, out_features))
        return fail(CATAN_EINVAL, "catan_linear_rows: bad arguments / unsupported widths (in multiple of 8 and <= 192, out <= 192, ceil(in/32)*ceil(out/16) <= 24)");
    if (mode < 0 || mode > 3 || (mode >= 2 && !aux) || (mode != 0 && (out_features & 7)))
        return fail(CATAN_EINVAL, "catan_linear_rows_fused: mode 0..3; modes 2, 3 need aux; a fused epilogue needs out_features % 8 == 0");
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)aux) & 15) return fail(CATAN_EINVAL, "catan_linear_rows: x, w, y, aux must be 16-byte aligned");
    const int ks = (in_features + 31) / 32, nt = (out_features + 15) / 16;
    switch (ks) {
    case 1: return linrows_dispatch_nt<1>(nt, x, w, bias, y, rows, in_features, out_features, aux, mode, S(stream));
    case 2: return linrows_dispatch_nt<2>(nt, x, w, bias, y, rows, in_features, out_features, aux, mode, S(stream));
    case 3: return linrows_dispatch_nt<3>(nt, x, w, bias, y, rows, in_features, out_features, aux, mode, S(stream));
    case 4: return linrows_dispatch_nt<4>(nt, x, w, bias, y, rows, in_features, out_features, aux, mode, S(stream));
    default: return linrows_dispatch_nt<6>(nt, x, w, bias, y, rows, in_features, out_features, aux, mode, S(stream));
    }
}

int catan_linear_rows(const void* x, const void* w, const void* bias, void* y, int64_t rows, int in_features, int out_features,
                      catan_stream_t stream) {
    return catan_linear_rows_fused(x, w, bias, y, rows, in_features, out_features, nullptr, 0, stream);
}

int catan_lstm_cell_fwd(const void* gx, const void* gh, const float* c_prev, const float* mask, float* h_out, float* c_out, int64_t rows,
                        int hidden, int is_bf16, catan_stream_t stream) {
    if (!gx || !gh || !c_prev || !h_out || !c_out || rows <= 0 || hidden <= 0 || (hidden & 3))
        return fail(CATAN_EINVAL, "catan_lstm_cell_fwd: bad arguments (hidden must be a multiple of 4)");
    if (((uintptr_t)gx | (uintptr_t)gh | (uintptr_t)c_prev | (uintptr_t)h_out | (uintptr_t)c_out) & 15)
        return fail(CATAN_EINVAL, "catan_lstm_cell_fwd: buffers must be 16-byte aligned");
    return is_bf16 ? lstm_cell_launch<__hip_bfloat16>(false, gx, gh, c_prev, mask, nullptr, nullptr, h_out, c_out, rows, hidden, S(stream))
                   : lstm_cell_launch<float>(false, gx, gh, c_prev, mask, nullptr, nullptr, h_out, c_out, rows, hidden, S(stream));
}

int catan_lstm_cell_bwd(const void* gx, const void* gh, const float* c_prev, const float* mask, const float* dh, const float* dc, void* dgates,
                        float* dc_prev, int64_t rows, int hidden, int is_bf16, catan_stream_t stream) {
    if (!gx || !gh || !c_prev || !dh || !dc || !dgates || !dc_prev || rows <= 0 || hidden <= 0 || (hidden & 3))
        return fail(CATAN_EINVAL, "catan_lstm_cell_bwd: bad arguments (hidden must be a multiple of 4)");
    if (((uintptr_t)gx | (uintptr_t)gh | (uintptr_t)c_prev | (uintptr_t)dh | (uintptr_t)dc | (uintptr_t)dgates | (uintptr_t)dc_prev) & 15)
        return fail(CATAN_EINVAL, "catan_lstm_cell_bwd: buffers must be 16-byte aligned");
    return is_bf16 ? lstm_cell_launch<__hip_bfloat16>(true, gx, gh, c_prev, mask, dh, dc, dgates, dc_prev, rows, hidden, S(stream))
                   : lstm_cell_launch<float>(true, gx, gh, c_prev, mask, dh, dc, dgates, dc_prev, rows, hidden, S(stream));
}

int catan_categorical_fwd(const float* logits, const float* mask, int64_t mask_ld, const int64_t* given, const float* u, int64_t* action,
                          float* logp, float* entropy, float* lse, int64_t rows, int K, catan_stream_t stream) {
    if (!logits || !mask || !action || !logp || !entropy || !lse || rows <= 0 || K <= 0 || mask_ld < K)
        return fail(CATAN_EINVAL, "catan_categorical_fwd: bad arguments");
    hipLaunchKernelGGL(k_categorical_fwd, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, S(stream), logits, mask, (long)mask_ld,
                       (const long long*)given, u, (long long*)action, logp, entropy, lse, (long)rows, K);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

int catan_categorical_bwd(const float* logits, const float* mask, int64_t mask_ld, const int64_t* action, const float* lse, const float* entropy,
                          const float* dlogp, const float* dent, float* dlogits, int64_t rows, int K, catan_stream_t stream) {
    if (!logits || !mask || !action || !lse || !entropy || !dlogp || !dent || !dlogits || rows <= 0 || K <= 0 || mask_ld < K)
        return fail(CATAN_EINVAL, "catan_categorical_bwd: bad arguments");
    hipLaunchKernelGGL(k_categorical_bwd, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, S(stream), logits, mask, (long)mask_ld,
                       (const long long*)action, lse, entropy, dlogp, dent, dlogits, (long)rows, K);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

static int cat_bits_args(CatBits& mb, const uint32_t* packed, int64_t pitch_words, const int64_t* rows_idx, const int32_t* segs, int64_t rows, int K) {
    if (!packed || !segs || pitch_words < 1 || rows <= 0 || K <= 0) return 1;
    mb.pm = packed; mb.pitch = (long)pitch_words; mb.rows = (const long long*)rows_idx;
    mb.n0 = segs[0]; mb.n1 = segs[1];
    for (int k = 0; k < 3; k++) {
        mb.off[k] = segs[2 + 2 * k]; mb.aoff[k] = segs[3 + 2 * k];
        if (mb.off[k] < 0 || mb.off[k] + K > pitch_words * 32 || mb.aoff[k] + K > pitch_words * 32) return 1;
    }
    return mb.n0 < 0 || mb.n1 < mb.n0;
}
int catan_categorical_bits_fwd(const float* logits, const uint32_t* packed, int64_t pitch_words, const int64_t* rows_idx, const int32_t* segs, const int64_t* given,
                               int64_t given_ld, int64_t* action, float* logp, float* entropy, float* lse, int64_t rows, int K, catan_stream_t stream) {
    CatBits mb;
    if (!logits || !action || !logp || !entropy || !lse || cat_bits_args(mb, packed, pitch_words, rows_idx, segs, rows, K) || (given && given_ld < 1))
        return fail(CATAN_EINVAL, "catan_categorical_bits_fwd: bad arguments (segs: n0, n1, then (bit offset, AND offset or -1) x 3, inside the packed row)");
    hipLaunchKernelGGL(k_categorical_bits_fwd, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, S(stream), logits, mb, (const long long*)given, (long)given_ld,
                       (long long*)action, logp, entropy, lse, (long)rows, K);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int catan_categorical_bits_bwd(const float* logits, const uint32_t* packed, int64_t pitch_words, const int64_t* rows_idx, const int32_t* segs, const int64_t* action,
                               const float* lse, const float* entropy, const float* dlogp, const float* dent, float* dlogits, int64_t rows, int K, catan_stream_t stream) {
    CatBits mb;
    if (!logits || !action || !lse || !entropy || !dlogp || !dent || !dlogits || cat_bits_args(mb, packed, pitch_words, rows_idx, segs, rows, K))
        return fail(CATAN_EINVAL, "catan_categorical_bits_bwd: bad arguments");
    hipLaunchKernelGGL(k_categorical_bits_bwd, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, S(stream), logits, mb, (const long long*)action, lse, entropy, dlogp, dent,
                       dlogits, (long)rows, K);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

int32_t catan_tile_encoder_weight_elems(void) { return TE_WTOTAL; }
int32_t catan_tile_encoder_vec_elems(void) { return TE_VTOTAL; }
int catan_tile_encoder_fwd(const void* tiles, const void* weights, const float* vecs, void* out, int64_t boards, catan_stream_t stream) {
    if (!tiles || !weights || !vecs || !out || boards <= 0) return fail(CATAN_EINVAL, "catan_tile_encoder_fwd: bad arguments");
    long nb = (boards + TE_G - 1) / TE_G;
    TeSaves sv;
    memset(&sv, 0, sizeof sv);
    hipLaunchKernelGGL(k_tile_encoder_fwd<0>, dim3((unsigned)nb), dim3(TE_THREADS), 0, S(stream), (const unsigned short*)tiles, (const unsigned short*)weights, vecs,
                       (unsigned short*)out, (long)boards, sv, (long)(TE_L * TE_OUT));
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int catan_tile_encoder_fwd_train(const void* tiles, const void* weights, const float* vecs, void* out, int64_t out_pitch, const catan_te_saves_t* saves,
                                 int64_t boards, catan_stream_t stream) {
    if (!tiles || !weights || !vecs || !out || !saves || boards <= 0 || out_pitch < TE_L * TE_OUT) return fail(CATAN_EINVAL, "catan_tile_encoder_fwd_train: bad arguments");
    static_assert(sizeof(catan_te_saves_t) == sizeof(TeSaves), "the header's struct is the kernel's");
    const void* const* ptrs = reinterpret_cast<const void* const*>(saves);
    for (size_t i = 0; i < sizeof(TeSaves) / sizeof(void*); i++) {
        const bool optional = (i >= offsetof(TeSaves, n1) / sizeof(void*) && i < offsetof(TeSaves, n1) / sizeof(void*) + 2) ||
                              (i >= offsetof(TeSaves, n2) / sizeof(void*) && i < offsetof(TeSaves, n2) / sizeof(void*) + 2) ||
                              (i >= offsetof(TeSaves, h) / sizeof(void*) && i < offsetof(TeSaves, h) / sizeof(void*) + 2);
        if ((!ptrs[i] && !optional) || ((uintptr_t)ptrs[i] & 15))
            return fail(CATAN_EINVAL, "catan_tile_encoder_fwd_train: every save buffer but n1 / n2 / h must be set, all 16-byte aligned");
    }
    TeSaves sv;
    memcpy(&sv, saves, sizeof sv);
    long nb = (boards + TE_G - 1) / TE_G;
    hipLaunchKernelGGL(k_tile_encoder_fwd<1>, dim3((unsigned)nb), dim3(TE_THREADS), 0, S(stream), (const unsigned short*)tiles, (const unsigned short*)weights, vecs,
                       (unsigned short*)out, (long)boards, sv, (long)out_pitch);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

int32_t catan_card_summary_params(void) { return CS_NPAR; }

static int card_summary_check(const void* ids, int esz, int64_t pitch, const int32_t* lens, const float* params, int64_t rows) {
    if (!ids || !lens || !params || rows <= 0 || pitch < 1 || (esz != 1 && esz != 4 && esz != 8)) return fail(CATAN_EINVAL, "catan_card_summary: bad arguments");
    return CATAN_OK;
}
int catan_card_summary_fwd(const void* ids, int id_bytes, int64_t pitch, const int32_t* lens, const float* params, float eps, float* out,
                           int32_t* keys, int64_t rows, catan_stream_t stream) {
    if (card_summary_check(ids, id_bytes, pitch, lens, params, rows) || !out) return fail(CATAN_EINVAL, "catan_card_summary_fwd: bad arguments");
    hipLaunchKernelGGL(k_card_summary_fwd, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, S(stream), ids, id_bytes, (long)pitch, lens, params, eps, out, (long)rows, keys);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int catan_card_summary_lookup(const void* ids, int id_bytes, int64_t pitch, const int32_t* lens, const float* table, const float* params, float eps,
                              float* out, int64_t rows, catan_stream_t stream) {
    if (card_summary_check(ids, id_bytes, pitch, lens, params, rows) || !out || !table) return fail(CATAN_EINVAL, "catan_card_summary_lookup: bad arguments");
    hipLaunchKernelGGL(k_card_summary_lookup, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, S(stream), ids, id_bytes, (long)pitch, lens, table, params, eps, out, (long)rows);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int catan_card_summary_bwd(const void* ids, int id_bytes, int64_t pitch, const int32_t* lens, const float* params, float eps, const float* dout,
                           float* dparams, const int32_t* only_unkeyed, const int32_t* n_unkeyed, int64_t rows, catan_stream_t stream) {
    if (card_summary_check(ids, id_bytes, pitch, lens, params, rows) || !dout || !dparams || (n_unkeyed && !only_unkeyed))
        return fail(CATAN_EINVAL, "catan_card_summary_bwd: bad arguments");
    const int rpl = rows >= 65536 ? 4 : 1;                 // lists per lane: few rows (the pattern table) want all the lanes they can get
    const int nb = (int)((rows + 256 * rpl - 1) / (256 * rpl));
    const int row_blocks = rows < 65536 ? nb : 0;          // ... and one query class per workgroup (CS_V x nb workgroups per slice)
    const dim3 grid((unsigned)(row_blocks > 0 ? CS_V * nb : nb), 7);
    hipLaunchKernelGGL(k_card_summary_bwd, grid, dim3(256), 0, S(stream), ids, id_bytes, (long)pitch, lens, params, eps, dout, dparams, (long)rows, only_unkeyed, rpl, n_unkeyed,
                       row_blocks);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}
int32_t catan_card_summary_patterns(void) { return CS_PATTERNS; }
int catan_card_pattern_sum(const int32_t* keys, const float* dout, float* dpat, int replicas, int32_t* n_unkeyed, int64_t rows, catan_stream_t stream) {
    if (!keys || !dout || !dpat || rows <= 0 || replicas < 1) return fail(CATAN_EINVAL, "catan_card_pattern_sum: bad arguments");
    hipLaunchKernelGGL(k_card_pattern_sum, dim3((unsigned)((rows + CPS_ROWS - 1) / CPS_ROWS)), dim3(256), 0, S(stream), keys, dout, dpat, (long)rows, replicas, n_unkeyed);
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

int64_t catan_missed_speculation_count(catan_env_t* e, catan_stream_t stream) {
    if (!e) return -1;
    u32 v = 0;
    if (hipMemcpyAsync(&v, e->err + 2, sizeof v, hipMemcpyDeviceToHost, S(stream)) != hipSuccess) return -1;
    if (hipStreamSynchronize(S(stream)) != hipSuccess) return -1;
    return (int64_t)v;
}

int catan_calib_copy(void* dst, const void* src, int64_t bytes, catan_stream_t stream) {
    if (!dst || !src || bytes <= 0 || bytes % 16) return fail(CATAN_EINVAL, "catan_calib_copy: bad arguments");
    hipLaunchKernelGGL(k_calib_copy, dim3(4096), dim3(BLOCK), 0, S(stream), (const uint4*)src, (uint4*)dst, (long)(bytes / 16));
    HIPCHK(hipGetLastError());
    return CATAN_OK;
}

int catan_profile_enable(catan_env_t* e, int on) {
    if (!e) return fail(CATAN_EINVAL, "catan_profile_enable: null handle");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemset(e->prof, 0, PROF_WORDS * sizeof(unsigned long long)));
    HIPCHK(hipMemset(e->prof_wave, 0, (size_t)(e->N / 16 + SORT_PAD_WAVES) * 8 * sizeof(u32)));
    e->prof_on = on;
    return CATAN_OK;
}
int catan_profile_read_waves(catan_env_t* e, uint32_t* out) {
    if (!e || !out) return fail(CATAN_EINVAL, "catan_profile_read_waves: bad arguments");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, e->prof_wave, (size_t)prof_wave_rows(e->N) * 8 * sizeof(u32), hipMemcpyDeviceToHost));
    return CATAN_OK;
}
int catan_profile_read(catan_env_t* e, uint64_t* out16) {
    if (!e || !out16) return fail(CATAN_EINVAL, "catan_profile_read: bad arguments");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out16, e->prof, PROF_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return CATAN_OK;
}

}  // extern "C"
