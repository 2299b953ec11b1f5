// catan_ppo.hip - fused GAE and clipped-PPO loss (forward + backward) kernels.
//
//   k_gae            BatchProcessor.compute_advantages_alt, the reverse-time recurrence of
//                    RL/ppo/process_batch.py:134-141 (one lane per game, coalesced [t][n] rows)
//   k_adv_stats/k_adv_normalise   the global normalisation of process_batch.py:142 (mean / unbiased std over T*N;
//                    exposed as (sum, sumsq, count) so that N>1 ranks can all-reduce three scalars - SURVEY 8(e))
//   k_ppo_loss       RL/ppo/ppo.py:46-48 (value normaliser) + :54-66 (loss) and its analytic gradient w.r.t.
//                    action_log_probs and values in the same pass
// fp32 recurrences are evaluated in the reference's operand order with contraction disabled (__fmul_rn/__fadd_rn)
// so returns match torch bit-for-bit; reductions accumulate in fp64 in a fixed order (deterministic).
#include <hip/hip_runtime.h>

namespace catan {

constexpr int GAE_BLOCK = 256;

// rewards [T][N], values [T+1][N] (denormalised), masks [T+1][N] -> returns [T][N], adv_raw [T][N];
// partial [2 * gridDim.x] = per-block (sum, sumsq) of adv_raw in fp64
__global__ __launch_bounds__(GAE_BLOCK) void k_gae(const float* __restrict__ rewards, const float* __restrict__ values,
                                                   const float* __restrict__ masks, long T, long N, float gamma, float gl,
                                                   float* __restrict__ returns, float* __restrict__ adv, double* __restrict__ partial) {
    __shared__ double sh[2][GAE_BLOCK];
    const long n = (long)blockIdx.x * GAE_BLOCK + threadIdx.x;
    double sum = 0.0, sumsq = 0.0;
    if (n < N) {
        float gae = 0.0f;
        float v1 = values[T * N + n];
        for (long t = T - 1; t >= 0; t--) {
            const float m1 = masks[(t + 1) * N + n], v0 = values[t * N + n], r = rewards[t * N + n];
            // delta = r + gamma * v1 * m1 - v0          (process_batch.py:137)
            const float delta = __fsub_rn(__fadd_rn(r, __fmul_rn(__fmul_rn(gamma, v1), m1)), v0);
            // gae = delta + gamma * lambda * m1 * gae   (process_batch.py:138)
            gae = __fadd_rn(delta, __fmul_rn(__fmul_rn(gl, m1), gae));
            const float ret = __fadd_rn(gae, v0);        // process_batch.py:139
            returns[t * N + n] = ret;
            const float a = __fsub_rn(ret, v0);          // process_batch.py:141
            adv[t * N + n] = a;
            sum += (double)a; sumsq += (double)a * (double)a;
            v1 = v0;
        }
    }
    sh[0][threadIdx.x] = sum; sh[1][threadIdx.x] = sumsq;
    __syncthreads();
    for (int s = GAE_BLOCK / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) { sh[0][threadIdx.x] += sh[0][threadIdx.x + s]; sh[1][threadIdx.x] += sh[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = sh[0][0]; partial[2 * blockIdx.x + 1] = sh[1][0]; }
}
// stats[3] = (sum, sumsq, count) from the per-block partials, fixed order
__global__ __launch_bounds__(GAE_BLOCK) void k_adv_stats(const double* __restrict__ partial, int nblocks, double count, double* __restrict__ stats) {
    __shared__ double sh[2][GAE_BLOCK];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += GAE_BLOCK) { a += partial[2 * i]; b += partial[2 * i + 1]; }
    sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b;
    __syncthreads();
    for (int s = GAE_BLOCK / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) { sh[0][threadIdx.x] += sh[0][threadIdx.x + s]; sh[1][threadIdx.x] += sh[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { stats[0] = sh[0][0]; stats[1] = sh[1][0]; stats[2] = count; }
}
// adv = (adv - mean) / (std_unbiased + 1e-5)   (process_batch.py:142); stats = GLOBAL (sum, sumsq, count)
__global__ __launch_bounds__(GAE_BLOCK) void k_adv_normalise(float* __restrict__ adv, long total, const double* __restrict__ stats) {
    const double cnt = stats[2], mean = stats[0] / cnt;
    const double var = (stats[1] - cnt * mean * mean) / (cnt - 1.0);
    const float fm = (float)mean, fs = (float)(sqrt(var > 0.0 ? var : 0.0)) + 1e-5f;
    for (long i = (long)blockIdx.x * GAE_BLOCK + threadIdx.x; i < total; i += (long)gridDim.x * GAE_BLOCK)
        adv[i] = (adv[i] - fm) / fs;
}

struct PpoArgs { float clip, value_coef, norm_mean, norm_std; int use_norm; };
// losses[0] = action loss, losses[1] = value loss (means over B = T*N/num_mini_batch rows, 2 000 .. ~2*10^5);
// d_logp / d_values = gradient of value_coef * L_v + L_pi (upstream gradient 1).  PPO_BLOCKS workgroups stream their share of
// the rows (24 B read + 8 B written per row); the per-workgroup fp64 partial sums go to `ws` and the workgroup that arrives
// last adds them up in index order, so the result does not depend on the schedule (ws: 2 * PPO_BLOCKS + 1 doubles, the last
// one the arrival counter - zero before the first call, left zero by every call).
constexpr int PPO_BLOCKS = 256, PPO_THREADS = 256;
static_assert(PPO_BLOCKS <= PPO_THREADS, "the last workgroup reduces one partial sum per lane");
__global__ __launch_bounds__(PPO_THREADS) void k_ppo_loss(const float* __restrict__ logp, const float* __restrict__ old_logp,
                                                          const float* __restrict__ adv, const float* __restrict__ values,
                                                          const float* __restrict__ old_values, const float* __restrict__ returns,
                                                          long B, PpoArgs a, float* __restrict__ losses,
                                                          float* __restrict__ d_logp, float* __restrict__ d_values, double* __restrict__ ws) {
    __shared__ double sh[2][PPO_THREADS];
    __shared__ bool last;
    double la = 0.0, lv = 0.0;
    const float invB = 1.0f / (float)B;
    const float lo = 1.0f - a.clip, hi = 1.0f + a.clip;
    for (long i = (long)blockIdx.x * PPO_THREADS + threadIdx.x; i < B; i += (long)gridDim.x * PPO_THREADS) {
        float vp = old_values[i], ret = returns[i];
        if (a.use_norm) { vp = (vp - a.norm_mean) / (a.norm_std + 1e-4f); ret = (ret - a.norm_mean) / (a.norm_std + 1e-4f); }   // ppo.py:46-48
        const float ratio = expf(logp[i] - old_logp[i]);                    // ppo.py:54
        const float ad = adv[i];
        const float s1 = ratio * ad;                                        // :55
        const float rc = fminf(fmaxf(ratio, lo), hi);
        const float s2 = rc * ad;                                           // :56
        la += (double)(-fminf(s1, s2));                                     // :57
        const bool inside = ratio >= lo && ratio <= hi;
        d_logp[i] = (s1 <= s2 || inside) ? -s1 * invB : 0.0f;               // d(-min)/dlogp = -ratio * adv on the active branch
        const float v = values[i];
        const float dv = v - vp;
        const float vc = vp + fminf(fmaxf(dv, -a.clip), a.clip);            // :59-60
        const float e1 = v - ret, e2 = vc - ret;
        const float l1 = e1 * e1, l2 = e2 * e2;                             // :61-62
        lv += 0.5 * (double)fmaxf(l1, l2);                                  // :63
        const bool vin = dv >= -a.clip && dv <= a.clip;
        d_values[i] = a.value_coef * invB * ((l1 >= l2) ? e1 : (vin ? e2 : 0.0f));
    }
    sh[0][threadIdx.x] = la; sh[1][threadIdx.x] = lv;
    __syncthreads();
    for (int s = PPO_THREADS / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) { sh[0][threadIdx.x] += sh[0][threadIdx.x + s]; sh[1][threadIdx.x] += sh[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ws[blockIdx.x] = sh[0][0]; ws[PPO_BLOCKS + blockIdx.x] = sh[1][0];
        __threadfence();
        unsigned long long* counter = reinterpret_cast<unsigned long long*>(ws + 2 * PPO_BLOCKS);
        last = atomicAdd(counter, 1ull) == (unsigned long long)gridDim.x - 1;
    }
    __syncthreads();
    if (last) {                                             // the whole last workgroup: partial k at lane k, then the same tree as above
        __threadfence();
        const bool have = threadIdx.x < gridDim.x;
        sh[0][threadIdx.x] = have ? __hip_atomic_load(&ws[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        sh[1][threadIdx.x] = have ? __hip_atomic_load(&ws[PPO_BLOCKS + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        __syncthreads();
        for (int s2 = PPO_THREADS / 2; s2 > 0; s2 >>= 1) {
            if (threadIdx.x < s2) { sh[0][threadIdx.x] += sh[0][threadIdx.x + s2]; sh[1][threadIdx.x] += sh[1][threadIdx.x + s2]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            losses[0] = (float)(sh[0][0] / (double)B); losses[1] = (float)(sh[1][0] / (double)B);
            *reinterpret_cast<unsigned long long*>(ws + 2 * PPO_BLOCKS) = 0ull;
        }
    }
}

}  // namespace catan
