// catan_rows.hip - row gathers of the PPO learner (RL/ppo/ppo.py:44-50: `obs_batch = [obs[i] for i in indices]`; here the rollout
// lives in HBM as (T + 1, N, 1 787) bf16 rows of 3 574 bytes - 2-byte aligned - and a minibatch is 204 800 of its rows):
//   k_gather_rows     dst[j] = src[idx[j]] for byte rows of any even length and alignment (the generic indexing kernel moves such rows
//                     element by element: 580 us for the 180 000 x 2 280 B tile features of a minibatch, 410 MB each way)
//   k_expand_rows16   out[j] = src[inv[j]] for rows that are whole 16-byte pieces (a per-board result spread to the rows showing it)
//   k_segment_sum16   its backward: out[u] = sum of dy[order[j]] over start[u] <= j < start[u + 1] (the rows of board u), fp32 sums
#pragma once

namespace catan {

__global__ __launch_bounds__(256) void k_gather_rows(const unsigned char* __restrict__ src, long src_pitch, const long long* __restrict__ idx, long n,
                                                     unsigned char* __restrict__ dst, long dst_pitch, int row_bytes) {
    const int words = row_bytes >> 2;                                   // whole 4-byte words; an odd 2-byte tail follows
    for (long r = blockIdx.x; r < n; r += gridDim.x) {
        const unsigned char* s = src + idx[r] * src_pitch;
        unsigned char* d = dst + r * dst_pitch;
        const bool s4 = (((unsigned long long)s) & 3) == 0, d4 = (((unsigned long long)d) & 3) == 0;
        for (int w = threadIdx.x; w < words; w += 256) {
            unsigned v;
            if (s4) v = *reinterpret_cast<const unsigned*>(s + 4 * w);
            else v = (unsigned)*reinterpret_cast<const unsigned short*>(s + 4 * w) | ((unsigned)*reinterpret_cast<const unsigned short*>(s + 4 * w + 2) << 16);
            if (d4) *reinterpret_cast<unsigned*>(d + 4 * w) = v;
            else { *reinterpret_cast<unsigned short*>(d + 4 * w) = (unsigned short)v; *reinterpret_cast<unsigned short*>(d + 4 * w + 2) = (unsigned short)(v >> 16); }
        }
        if ((row_bytes & 2) && threadIdx.x == 0) *reinterpret_cast<unsigned short*>(d + 4 * words) = *reinterpret_cast<const unsigned short*>(s + 4 * words);
    }
}

__global__ __launch_bounds__(256) void k_expand_rows16(const uint4* __restrict__ src, const long long* __restrict__ inv, long n, uint4* __restrict__ out, int chunks) {
    const long total = n * chunks;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / chunks; const int c = (int)(e - r * chunks);
        out[e] = src[inv[r] * chunks + c];
    }
}

__global__ __launch_bounds__(256) void k_segment_sum16(const uint4* __restrict__ dy, const long long* __restrict__ order, const long long* __restrict__ start, long U,
                                                       uint4* __restrict__ out, int chunks, long dy_pitch) {      // dy_pitch: 16-byte pieces between two rows of dy
    const long total = U * chunks;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long u = e / chunks; const int c = (int)(e - u * chunks);
        const long j0 = start[u], j1 = start[u + 1];
        float acc[8] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
        for (long j = j0; j < j1; j++) {
            const uint4 v = dy[order[j] * dy_pitch + c];
            acc[0] += __uint_as_float(v.x << 16); acc[1] += __uint_as_float(v.x & 0xFFFF0000u);
            acc[2] += __uint_as_float(v.y << 16); acc[3] += __uint_as_float(v.y & 0xFFFF0000u);
            acc[4] += __uint_as_float(v.z << 16); acc[5] += __uint_as_float(v.z & 0xFFFF0000u);
            acc[6] += __uint_as_float(v.w << 16); acc[7] += __uint_as_float(v.w & 0xFFFF0000u);
        }
        uint4 o;
        if (j1 - j0 == 1) o = dy[order[j0] * dy_pitch + c];                  // (one row: its bits, not a round trip through fp32 - the same value)
        else {
            o = make_uint4(pk_bf(acc[0], acc[1]), pk_bf(acc[2], acc[3]), pk_bf(acc[4], acc[5]), pk_bf(acc[6], acc[7]));
        }
        out[e] = o;
    }
}

// out[r] = src_0[r] | src_1[r] | ... (rows of whole 16-byte pieces, out's rows out_chunks pieces apart): the trunk input of the policy
// net (observation_module.py:58-60 concatenates the tile encoding and the player modules' outputs) - torch's cat moves these 1 984-byte
// rows at 2.5 TB/s (0.33 ms per 204 800 rows)
constexpr int CC_MAX = 4;
struct ConcatSrc { int n; const uint4* src[CC_MAX]; int chunks[CC_MAX]; int first[CC_MAX]; };
__global__ __launch_bounds__(256) void k_concat_rows16(ConcatSrc cs, uint4* __restrict__ out, long rows, int row_chunks, int out_chunks) {
    const long total = rows * row_chunks;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / row_chunks; const int c = (int)(e - r * row_chunks);
        int k = 0;
#pragma unroll
        for (int j = 1; j < CC_MAX; j++) if (j < cs.n && c >= cs.first[j]) k = j;
        out[r * out_chunks + c] = cs.src[k][r * cs.chunks[k] + (c - cs.first[k])];
    }
}

// The backward of a gather whose index list is a concatenation of RANGES of a permutation (the heads' rows of a minibatch: perm = the rows
// sorted by action type, every head takes one or two contiguous runs of it, a row appears in at most three lists): out row perm[p] = the
// sum (fp32, rounded to bf16) of the rows dy[off_k + p - a_k] over the ranges a_k <= p < b_k (+ add0 / add1 rows perm[p]), zeros for a row
// with no term.  Autograd's
// index_put sorts the 2 x 10^5 indices again in every step (0.32 ms); this is one pass over dy and out.
constexpr int SR_MAX = 16;
struct ScatterRanges { int n; long a[SR_MAX], b[SR_MAX], off[SR_MAX]; };
__global__ __launch_bounds__(256) void k_scatter_ranges16(const uint4* __restrict__ dy, long dy_pitch, const long long* __restrict__ perm, long n_perm,
                                                          ScatterRanges rg, const uint4* __restrict__ add0, const uint4* __restrict__ add1,
                                                          uint4* __restrict__ out, int chunks) {
    const long total = n_perm * chunks;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long p = e / chunks; const int c = (int)(e - p * chunks);
        float acc[8] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
        uint4 one = make_uint4(0, 0, 0, 0);
        int hits = 0;
        for (int k = 0; k < rg.n; k++) {
            if (p < rg.a[k] || p >= rg.b[k]) continue;
            const uint4 v = dy[(rg.off[k] + p - rg.a[k]) * dy_pitch + c];
            one = v; hits++;
            acc[0] += __uint_as_float(v.x << 16); acc[1] += __uint_as_float(v.x & 0xFFFF0000u);
            acc[2] += __uint_as_float(v.y << 16); acc[3] += __uint_as_float(v.y & 0xFFFF0000u);
            acc[4] += __uint_as_float(v.z << 16); acc[5] += __uint_as_float(v.z & 0xFFFF0000u);
            acc[6] += __uint_as_float(v.w << 16); acc[7] += __uint_as_float(v.w & 0xFFFF0000u);
        }
        // add0 / add1 (optional): further gradients of the SAME rows (the source tensor's other consumers), [n_perm][chunks] like out
        const long o = perm[p] * chunks + c;
        if (add0) { const uint4 v = add0[o]; one = v; hits++;
            acc[0] += __uint_as_float(v.x << 16); acc[1] += __uint_as_float(v.x & 0xFFFF0000u); acc[2] += __uint_as_float(v.y << 16); acc[3] += __uint_as_float(v.y & 0xFFFF0000u);
            acc[4] += __uint_as_float(v.z << 16); acc[5] += __uint_as_float(v.z & 0xFFFF0000u); acc[6] += __uint_as_float(v.w << 16); acc[7] += __uint_as_float(v.w & 0xFFFF0000u); }
        if (add1) { const uint4 v = add1[o]; one = v; hits++;
            acc[0] += __uint_as_float(v.x << 16); acc[1] += __uint_as_float(v.x & 0xFFFF0000u); acc[2] += __uint_as_float(v.y << 16); acc[3] += __uint_as_float(v.y & 0xFFFF0000u);
            acc[4] += __uint_as_float(v.z << 16); acc[5] += __uint_as_float(v.z & 0xFFFF0000u); acc[6] += __uint_as_float(v.w << 16); acc[7] += __uint_as_float(v.w & 0xFFFF0000u); }
        // (one row: its bits, not a round trip through fp32 - the same value)
        out[o] = hits == 1 ? one : make_uint4(pk_bf(acc[0], acc[1]), pk_bf(acc[2], acc[3]), pk_bf(acc[4], acc[5]), pk_bf(acc[6], acc[7]));
    }
}

}  // namespace catan

namespace catan {
// k_weight_images: every derived image of the net's fp32 parameters that a training step reads - bf16 copies, transposed bf16
// copies, the fused tile encoder's packed weight / vector blocks - refreshed by ONE launch after the optimiser step.  As separate
// casts / transposes / pads they were ~500 launches of 2-4 us per step (1.9 of its 34 ms on the device, and as much host time).
// One table row per image: a strided 2-D copy with conversion, dst[r * d_r + c * d_c] = conv(src[r * s_r + c * s_c]).
struct WeightImage { const float* src; void* dst; int rows, cols; long s_r, s_c, d_r, d_c; int mode; int pad_; };   // mode 0: -> bf16, 1: -> fp32, 2: -> bf16 -> fp32
__global__ __launch_bounds__(256) void k_weight_images(const WeightImage* __restrict__ table, int n) {
    const WeightImage w = table[blockIdx.x];
    const long total = (long)w.rows * w.cols;
    for (long e = (long)blockIdx.y * 256 + threadIdx.x; e < total; e += (long)gridDim.y * 256) {
        const long r = e / w.cols, c = e - r * w.cols;
        const float v = w.src[r * w.s_r + c * w.s_c];
        const long o = r * w.d_r + c * w.d_c;
        if (w.mode == 0) reinterpret_cast<unsigned short*>(w.dst)[o] = te_to_bf(v);
        else if (w.mode == 1) reinterpret_cast<float*>(w.dst)[o] = v;
        else reinterpret_cast<float*>(w.dst)[o] = te_bf(te_to_bf(v));
    }
}
// The conditioning columns, masks and step weights of a recurrent resource head evaluated for GIVEN picks (the PPO update;
// RL/models/action_heads_module.py:258-329: with the four picks known, every step's inputs are functions of the earlier picks alone).
// One lane per row b; step-major outputs (row i * B + b = step i of row b), as policy._recurrent_given lays them out for ONE
// evaluation of the head over 4 B rows:
//   cond [4 B][kf + 6]  fixed[b][0..kf) | out_i[b][0..6): how often each resource was picked before step i, column 0 ("stop") cleared
//   mask [4 B][6]       from_hand: (hand - earlier picks, clamped at 0) > 0, else ones; column 0: step 0: the hand is empty, later steps: 1
//   given [4 B]         the picks, step-major;   keep [B][4]: 1, pick_0 > 0, pick_1 > 0, pick_2 > 0
//   out_final [B][6]    the counts of all four picks, column 0 cleared
// acts: int64 rows acts_ld elements apart (a column window of the action rows); cur_res: float [B][6]; fixed: float [B][kf] or null.
template <bool BF16>
__global__ __launch_bounds__(256) void k_recurrent_given(const long long* __restrict__ acts, long acts_ld, const float* __restrict__ cur_res, const float* __restrict__ fixed,
                                                         int kf, int from_hand, long B, void* __restrict__ cond, float* __restrict__ mask, long long* __restrict__ given,
                                                         float* __restrict__ keep, float* __restrict__ out_final) {
    const long b = (long)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    float hand[6], cnt[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, tot = 0.f;
#pragma unroll
    for (int r = 0; r < 6; r++) { hand[r] = cur_res[b * 6 + r]; tot += hand[r]; }
    const int W = kf + 6;
    int prev = 1;
    for (int i = 0; i < 4; i++) {
        const long row = (long)i * B + b;
        long long a = acts[b * acts_ld + i];
        a = a < 0 ? 0 : (a > 5 ? 5 : a);
        given[row] = a;
        keep[b * 4 + i] = prev > 0 ? 1.f : 0.f;
        prev = (int)a;
#pragma unroll
        for (int r = 0; r < 6; r++) {
            float m = from_hand ? (fmaxf(hand[r] - cnt[r], 0.f) > 0.f ? 1.f : 0.f) : 1.f;
            if (r == 0) m = i == 0 ? (tot == 0.f ? 1.f : 0.f) : 1.f;
            mask[row * 6 + r] = m;
            const float o = r == 0 ? 0.f : cnt[r];
            if (BF16) reinterpret_cast<__hip_bfloat16*>(cond)[row * W + kf + r] = __float2bfloat16(o);
            else reinterpret_cast<float*>(cond)[row * W + kf + r] = o;
        }
        for (int k = 0; k < kf; k++) {
            const float f = fixed[b * kf + k];
            if (BF16) reinterpret_cast<__hip_bfloat16*>(cond)[row * W + k] = __float2bfloat16(f);
            else reinterpret_cast<float*>(cond)[row * W + k] = f;
        }
#pragma unroll
        for (int r = 0; r < 6; r++) cnt[r] += (r == (int)a) ? 1.f : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 6; r++) out_final[b * 6 + r] = r == 0 ? 0.f : cnt[r];
}

}  // namespace catan
