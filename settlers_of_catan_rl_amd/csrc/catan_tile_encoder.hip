// catan_tile_encoder.hip - the tile encoder of the policy net (RL/models/tile_encoder.py:41-91: Linear(60, 64) + LayerNorm +
// ReLU, two pre-norm transformer layers (4 heads x 16, FFN x 2), Linear(64, 25) + LayerNorm + ReLU per tile) as ONE forward
// kernel for inference (acting in rollouts and evaluation, the value passes of a PPO update, forward search).
//
// Unfused, every op of the encoder streams a [boards x 19, 64..192] activation tensor through HBM: ~14 GB per layer at
// 204 800 boards, 7.7 ms for the forward.  Here a workgroup (4 waves) takes 5 boards = 95 tokens through the whole encoder
// in LDS (73 KB: two workgroups per CU, so that one computes while the other waits for L2): HBM sees the 2 280 B of tile
// features per board once and 950 B of output; the 142 KB of weights come from L2 (29 KB per board), the 6.8 KB of bias /
// LayerNorm vectors are copied to LDS once per workgroup.  All per-token linear layers run on v_mfma_f32_16x16x32_bf16: the 95
// tokens are 6 row tiles of 16 (one padding row); a wave owns every fourth 16-column tile of a layer's output, holds that
// tile's weight fragments in registers - fetched one phase ahead of the product that uses them - and walks the row tiles
// (fragments = 16-byte LDS reads).  The product is formed transposed (weights as the A operand) so a lane ends up with 4
// consecutive columns of one token: 8-byte epilogue accesses.  LayerNorm is two lanes per token, the 19 x 19 attention one
// (board, head) per wave pass on v_mfma_f32_32x32x16_bf16 (S^T = K Q^T, in-lane softmax, O^T = V^T P^T).
// Measured (MI355X, 204 800 boards): 2.4 ms (8 boards per 512-thread workgroup, one workgroup per CU: 2.7 ms; SQ counters put
// the VALU floor of that instruction stream at ~1.35 ms, 60 % of its wave-cycles were waits on L2: tile staging, weights).
// Numerics: bf16 storage between the ops (as the unfused bf16-autocast path stores them), fp32 accumulation, fp32 softmax
// and LayerNorm statistics.
#pragma once

namespace catan {

constexpr int TE_G = 5, TE_L = 19, TE_TOK = TE_G * TE_L, TE_MT = (TE_TOK + 15) / 16, TE_ROWS = TE_MT * 16;    // 95 tokens, 6 tiles, 96 rows
constexpr int TE_THREADS = 256, TE_W = TE_THREADS / 64, TE_MP = TE_W / 4;    // waves per workgroup; row partitions of a product
constexpr int TE_LT = TE_THREADS >= 4 * TE_TOK ? 4 : 2;                       // lanes per token in a LayerNorm
constexpr int TE_IN = 60, TE_D = 64, TE_F = 128, TE_OUT = 25, TE_H = 4, TE_HD = 16;
constexpr int TE_PX = TE_D + 8, TE_PQ = 3 * TE_D + 8, TE_PH = TE_F + 8;      // LDS row pitches (bf16 elements; 16 B aligned rows)
// packed weights (bf16), row-major [N][K], K padded to a multiple of 32, N to a multiple of 16
constexpr int TE_W0 = 0;                                        // [64][64]  (first layer, K 60 -> 64)
constexpr int TE_WL = TE_W0 + 64 * 64;                          // per layer: Wqkv [192][64], Wo [64][64], W1 [128][64], W2 [64][128]
constexpr int TE_WL_SIZE = 192 * 64 + 64 * 64 + 128 * 64 + 64 * 128;
constexpr int TE_WP = TE_WL + 2 * TE_WL_SIZE;                   // [32][64]  (output projection, N 25 -> 32)
constexpr int TE_WTOTAL = TE_WP + 32 * 64;
// packed fp32 vectors: b0[64] ln0w[64] ln0b[64]; per layer: ln1w ln1b [64] bqkv[192] bo[64] ln2w ln2b [64] b1[128] b2[64]; bp[32] lnpw[32] lnpb[32]
constexpr int TE_V0 = 0, TE_VL = 192, TE_VL_SIZE = 64 * 2 + 192 + 64 + 64 * 2 + 128 + 64, TE_VP = TE_VL + 2 * TE_VL_SIZE, TE_VTOTAL = TE_VP + 96;

DEVI float te_bf(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
DEVI unsigned short te_to_bf(float v) { const __hip_bfloat16 h = __float2bfloat16(v); return *reinterpret_cast<const unsigned short*>(&h); }
// 8 consecutive bf16 of an LDS row (16-byte aligned) as floats
DEVI void te_load8(const unsigned short* p, float* o) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xFFFF0000u);
    o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xFFFF0000u);
    o[4] = __uint_as_float(u.z << 16); o[5] = __uint_as_float(u.z & 0xFFFF0000u);
    o[6] = __uint_as_float(u.w << 16); o[7] = __uint_as_float(u.w & 0xFFFF0000u);
}
DEVI void te_store8(unsigned short* p, const float* v) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pk_bf(v[0], v[1]), pk_bf(v[2], v[3]), pk_bf(v[4], v[5]), pk_bf(v[6], v[7]));
}

// the weight fragments (and bias values) of this wave's column tiles nt = wave, wave + 4, ... of one layer: fetched from L2 one
// phase AHEAD of the product that uses them (the ~1 us of a dependent global load would otherwise be paid 16 times per group)
template <int K, int N> struct TeW { bf16x8_t f[(N / 16 + 3) / 4][K / 32]; };
template <int K, int N>
DEVI void te_fetch(TeW<K, N>& w, const unsigned short* __restrict__ W, int lane, int wave) {
    const int lr = lane & 15, lk = (lane >> 4) * 8, ntb = wave / TE_MP;
#pragma unroll
    for (int i = 0; i < (N / 16 + 3) / 4; i++) {
        const int nt = ntb + 4 * i;
        if (nt < N / 16) {
#pragma unroll
            for (int ks = 0; ks < K / 32; ks++) {
                const uint4 u = *reinterpret_cast<const uint4*>(W + (long)(nt * 16 + lr) * K + ks * 32 + lk);
                w.f[i][ks] = *reinterpret_cast<const bf16x8_t*>(&u);
            }
        }
    }
}
// out[m][n] (+)= sum_k A[m][k] * W[n][k] + bias[n] for the TE_ROWS rows in LDS; this wave's column tiles nt = wave, wave + 4, ...
// MODE 0: store; 1: ReLU then store; 2: add to what `out` holds (residual stream)
template <int K, int N, int MODE>
DEVI void te_gemm(const unsigned short* A, int pa, const TeW<K, N>& w, const float* bias, unsigned short* out, int po, int lane, int wave) {
    constexpr int KS = K / 32, NT = N / 16;
    const int lr = lane & 15, lk = (lane >> 4) * 8, ntb = wave / TE_MP, part = wave % TE_MP;
    static_assert(TE_MT % TE_MP == 0, "equal row partitions");
    const int m0 = part * (TE_MT / TE_MP);
#pragma unroll
    for (int i = 0; i < (NT + 3) / 4; i++) {
        const int nt = ntb + 4 * i;
        if (nt >= NT) break;
        const float4 bv = *reinterpret_cast<const float4*>(bias + nt * 16 + 4 * (lane >> 4));   // (LDS)
#pragma unroll
        for (int mj = 0; mj < TE_MT / TE_MP; mj++) {
            const int mt = m0 + mj;
            // the product is formed TRANSPOSED (weights as the A operand): the lane holds token mt * 16 + (lane & 15) and the four
            // consecutive output columns nt * 16 + 4 * (lane >> 4) + r - one 8-byte LDS access per tile instead of four 2-byte ones
            unsigned short* o = out + (mt * 16 + lr) * po + nt * 16 + 4 * (lane >> 4);
            // the bias (MODE 2: + the residual) is the accumulator's initial value: the first MFMA takes it as its C operand
            f32x4_t acc = f32x4_t{bv.x, bv.y, bv.z, bv.w};
            if (MODE == 2) {
                const uint2 old = *reinterpret_cast<const uint2*>(o);
                acc[0] += __uint_as_float(old.x << 16); acc[1] += __uint_as_float(old.x & 0xFFFF0000u);
                acc[2] += __uint_as_float(old.y << 16); acc[3] += __uint_as_float(old.y & 0xFFFF0000u);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const uint4 u = *reinterpret_cast<const uint4*>(A + (mt * 16 + lr) * pa + ks * 32 + lk);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.f[i][ks], *reinterpret_cast<const bf16x8_t*>(&u), acc, 0, 0, 0);
            }
            float v[4] = { acc[0], acc[1], acc[2], acc[3] };
            if (MODE == 1) {
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], 0.f);
            }
            *reinterpret_cast<uint2*>(o) = make_uint2(pk_bf(v[0], v[1]), pk_bf(v[2], v[3]));
        }
    }
}
// LayerNorm over the first D columns of every token row (TE_LT adjacent lanes per token), optional ReLU; src / dst may alias
template <int D, bool RELU>
DEVI void te_layer_norm(const unsigned short* src, int ps, unsigned short* dst, int pd, const float* w, const float* b, int tid) {
    constexpr int DP = (D + 7) & ~7, E = DP / TE_LT;                  // rows are read / written in 16-byte pieces (the pad columns exist)
    static_assert(E % 8 == 0 && TE_TOK * TE_LT <= TE_THREADS, "one pass");
    const int t = tid / TE_LT, e0 = (tid % TE_LT) * E;
    if (t >= TE_TOK) return;
    float x[E], mean = 0.f;
#pragma unroll
    for (int i = 0; i < E; i += 8) te_load8(src + t * ps + e0 + i, x + i);
#pragma unroll
    for (int i = 0; i < E; i++) if (e0 + i < D || D == DP) mean += x[i];
#pragma unroll
    for (int m = 1; m < TE_LT; m <<= 1) mean += __shfl_xor(mean, m);
    mean *= 1.f / D;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < E; i++) { const float c = x[i] - mean; if (e0 + i < D || D == DP) var += c * c; }
#pragma unroll
    for (int m = 1; m < TE_LT; m <<= 1) var += __shfl_xor(var, m);
    const float rstd = rsqrtf(var * (1.f / D) + 1e-5f);
    const float4* w4 = reinterpret_cast<const float4*>(w + e0);       // (the packed vectors are padded to whole 16-byte pieces)
    const float4* b4 = reinterpret_cast<const float4*>(b + e0);
#pragma unroll
    for (int i = 0; i < E; i += 4) {
        const float4 wv = w4[i >> 2], bv = b4[i >> 2];
        const float ww[4] = { wv.x, wv.y, wv.z, wv.w }, bb[4] = { bv.x, bv.y, bv.z, bv.w };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float y = (e0 + i + k < D || D == DP) ? (x[i + k] - mean) * rstd * ww[k] + bb[k] : 0.f;
            if (RELU) y = fmaxf(y, 0.f);
            x[i + k] = y;
        }
    }
#pragma unroll
    for (int i = 0; i < E; i += 8) te_store8(dst + t * pd + e0 + i, x + i);
}
// The final LayerNorm(25) + ReLU, written straight to the output rows (token t of the group -> board t / 19, columns 25 (t % 19) ..):
// the rows are 50 bytes long and start on 2-byte boundaries, so the stores are 2 bytes wide whoever issues them - from here they
// cost no LDS round trip, no barrier and no index arithmetic per element.
DEVI void te_final_norm(const unsigned short* src, int ps, const float* w, const float* b, unsigned short* __restrict__ out, long out_pitch, int ntok, int tid) {
    constexpr int D = TE_OUT, DP = (D + 7) & ~7, E = DP / TE_LT;
    static_assert(E % 8 == 0 && TE_TOK * TE_LT <= TE_THREADS, "one pass");
    const int t = tid / TE_LT, e0 = (tid % TE_LT) * E;
    if (t >= ntok) return;
    float x[E], mean = 0.f;
#pragma unroll
    for (int i = 0; i < E; i += 8) te_load8(src + t * ps + e0 + i, x + i);
#pragma unroll
    for (int i = 0; i < E; i++) if (e0 + i < D) mean += x[i];
#pragma unroll
    for (int m = 1; m < TE_LT; m <<= 1) mean += __shfl_xor(mean, m);
    mean *= 1.f / D;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < E; i++) { const float c = x[i] - mean; if (e0 + i < D) var += c * c; }
#pragma unroll
    for (int m = 1; m < TE_LT; m <<= 1) var += __shfl_xor(var, m);
    const float rstd = rsqrtf(var * (1.f / D) + 1e-5f);
    const int g = t / TE_L;
    unsigned short* o = out + g * out_pitch + (t - g * TE_L) * D + e0;
    const float4* w4 = reinterpret_cast<const float4*>(w + e0);
    const float4* b4 = reinterpret_cast<const float4*>(b + e0);
#pragma unroll
    for (int i = 0; i < E; i += 4) {
        const float4 wv = w4[i >> 2], bv = b4[i >> 2];
        const float ww[4] = { wv.x, wv.y, wv.z, wv.w }, bb[4] = { bv.x, bv.y, bv.z, bv.w };
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (e0 + i + k < D) o[i + k] = te_to_bf(fmaxf((x[i + k] - mean) * rstd * ww[k] + bb[k], 0.f));
    }
}
// 4-head attention inside every board: qkv rows [token][3][head][16] -> out rows [token][head * 16 + d].  One (board, head) per
// wave pass, on v_mfma_f32_32x32x16_bf16 exactly as k_attn_mfma_fwd (catan_nn.hip) does it: S^T = K Q^T lands with lane = query
// column and the keys down the registers, so the softmax is in-lane plus one xor-32 exchange, and the probabilities are already
// the B operand of O^T = V^T P^T; the V^T fragments are gathered from the token-major V rows (2-byte LDS reads).
DEVI void te_attention(const unsigned short* qkv, unsigned short* out, int lane, int wave) {
    const int hf = lane >> 5, c31 = lane & 31;
    const bool rowok = c31 < TE_L;
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // V^T fragments through gfx950's transposing LDS read (ld_gather_tr, catan_nn.hip): a 16-lane group hands the instruction rows
    // base .. base + 3 of the token-major V slice and lane i gets dim i of those four keys.  Keys 0..15 are rows of this board; of
    // keys 16..31 only 16..18 are: the others carry probability 0 exactly and must only be FINITE, so their rows are clamped to the
    // board's last token (rows past it belong to the next board - or, for the group's last board, to whatever follows the array).
    const int i15 = lane & 15;
    const int vrow0 = 4 * hf + (i15 >> 2), vcol = (i15 & 3) * 4;
    const int vrow1 = 16 + vrow0 < TE_L ? 16 + vrow0 : TE_L - 1;
    for (int bh = wave; bh < TE_G * TE_H; bh += TE_W) {
        const int g = bh >> 2, h = bh & 3;
        const unsigned short* base = qkv + g * TE_L * TE_PQ + h * TE_HD;
        const bf16x8_t ka = ld_frag<TE_HD>(base + c31 * TE_PQ + TE_D, hf, rowok);
        const bf16x8_t qb = ld_frag<TE_HD>(base + c31 * TE_PQ, hf, rowok);
        const unsigned short* vb = base + 2 * TE_D + vcol;
        union { bf16x8_t f; wg_s4 q[2]; } va[2];
        va[0].q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s4*)(vb + vrow0 * TE_PQ));
        va[0].q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s4*)(vb + (vrow0 + 8) * TE_PQ));
        va[1].q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s4*)(vb + vrow1 * TE_PQ));
        va[1].q[1] = va[1].q[0];                                                   // keys 24..31: probability 0
        const f32x16_t st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qb, zero16, 0, 0, 0);     // S^T[j][i]
        constexpr int NR = TE_L <= 24 ? 12 : 16;                                   // registers 12..15 = keys 24..31: never valid
        float p[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int j = (r & 3) + 8 * (r >> 2) + 4 * hf;
            p[r] = j < TE_L ? st[r] : -INFINITY;
            mx = fmaxf(mx, p[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        // exp((s - max) / sqrt(16)) as one fma and v_exp_f32 (2^x); the 1 / sum (v_rcp_f32) goes onto the 8 output registers, not the 12 p
        constexpr float C = 0.25f * 1.44269504088896340736f;
        const float mc = -mx * C;
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < NR; r++) { p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(p[r], C, mc)); sum += p[r]; }
        sum += __shfl_xor(sum, 32);
        const float inv = __builtin_amdgcn_rcpf(sum);
        f32x16_t ot = zero16;
#pragma unroll
        for (int s = 0; s < 2; s++)
            ot = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c31 < TE_HD ? va[s].f : zero_bf8(), pack_bf8(p + 8 * s), ot, 0, 0, 0);   // O^T[d][i]
#pragma unroll
        for (int r = 0; r < 8; r++) ot[r] *= inv;
        if (rowok) st_head<TE_HD>(out + (g * TE_L + c31) * TE_PX + h * TE_HD, ot, hf);
    }
}

// Training (SAVE): the same forward also leaves, per token row, every activation the backward kernels of the sub-layers read
// (csrc/catan_nn.hip: LayerNorm, attention, row products, weight gradients) - the tensors the unfused training forward keeps for
// autograd, written once from LDS as 16-byte row pieces while the next phase computes: bf16 [boards * 19][width] each.
struct TeSaves {
    unsigned short* tiles64;          // [64]: the tile features, zero-padded 60 -> 64 (the first layer's input)
    unsigned short* a0;               // [64]: first_layer output (before its LayerNorm + ReLU)
    unsigned short* xin[2];           // [64]: the layer's input (residual stream)
    unsigned short* n1[2];            // [64]: LayerNorm 1 output (the QKV product's input); may be null: k_qkv_bwd_w<true> recomputes it from xin
    unsigned short* qkv[2];           // [192]
    unsigned short* o[2];             // [64]: attention output (the out-projection's input)
    unsigned short* xmid[2];          // [64]: residual stream after the attention sub-layer
    unsigned short* n2[2];            // [64]: LayerNorm 2 output (the FFN's input); may be null: k_ffn_bwd_w<., true> recomputes it from xmid
    unsigned short* h[2];             // [128]: relu(linear1); may be null: k_ffn_bwd_w<., true, true> recomputes it from the recomputed n2
    unsigned short* xfin;             // [64]: the last layer's output (out_proj's input)
    unsigned short* p;                // [25]: out_proj output (before the final LayerNorm + ReLU)
};
template <int W>
DEVI void te_dump(const unsigned short* lds, int pitch, unsigned short* __restrict__ g, long row0, int rows, int tid) {
    constexpr int CH = W / 8;
    g += row0 * W;
    for (int c = tid; c < rows * CH; c += TE_THREADS) {
        const int row = c / CH, ch = c - row * CH;
        *reinterpret_cast<uint4*>(g + (long)row * W + ch * 8) = *reinterpret_cast<const uint4*>(lds + row * pitch + ch * 8);
    }
}
// tiles: bf16 [boards][19][60] contiguous (8-byte aligned); out: bf16 [boards][19 * 25]; wts / vecs: the packed parameters
// SAVE_MODE 0: inference; 1: every activation of TeSaves (the sub-layer backward kernels of catan_te_bwd.hip)
template <int SAVE_MODE>
__global__ __launch_bounds__(TE_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_tile_encoder_fwd(const unsigned short* __restrict__ tiles, const unsigned short* __restrict__ wts,
                                                          const float* __restrict__ vecs, unsigned short* __restrict__ out, long boards, TeSaves sv, long out_pitch) {
    __shared__ __attribute__((aligned(16))) unsigned short X[TE_ROWS * TE_PX];     // residual stream
    __shared__ __attribute__((aligned(16))) unsigned short Nb[TE_ROWS * TE_PX];    // LayerNorm output / attention output / staged input
    __shared__ __attribute__((aligned(16))) unsigned short Q[TE_ROWS * TE_PQ];     // Q | K | V; the FFN hidden layer; the output projection
    __shared__ __attribute__((aligned(16))) float V[TE_VTOTAL];                    // every bias / LayerNorm vector (read in every phase)
    constexpr bool SAVE = SAVE_MODE == 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* const gvecs = vecs;
    for (int c = tid; c < TE_VTOTAL / 4; c += TE_THREADS) reinterpret_cast<float4*>(V)[c] = reinterpret_cast<const float4*>(gvecs)[c];
    {                                                                          // one group of TE_G boards per workgroup
        const long g0 = (long)blockIdx.x * TE_G;
        const int nb = (int)(boards - g0 < TE_G ? boards - g0 : TE_G);
        TeW<64, 64> w0; te_fetch<64, 64>(w0, wts + TE_W0, lane, wave);
        __syncthreads();
        // ---- stage the tile features: token rows of 60 bf16 (15 x 8 B), columns 60..63 and the rows beyond the last token zero
        {
            constexpr int IT = TE_ROWS * 16 / TE_THREADS;                         // 16 chunks of 4 elements per row; all loads in flight together
            static_assert(TE_ROWS * 16 % TE_THREADS == 0, "whole passes");
            uint2 v[IT];
#pragma unroll
            for (int i = 0; i < IT; i++) {
                const int c = tid + i * TE_THREADS, row = c >> 4, ch = c & 15;
                v[i] = make_uint2(0u, 0u);
                if (ch < 15 && row < nb * TE_L) v[i] = *reinterpret_cast<const uint2*>(tiles + (g0 * TE_L + row) * TE_IN + ch * 4);
            }
#pragma unroll
            for (int i = 0; i < IT; i++) {
                const int c = tid + i * TE_THREADS, row = c >> 4, ch = c & 15;
                *reinterpret_cast<uint2*>(Nb + row * TE_PX + ch * 4) = v[i];
            }
        }
        __syncthreads();
        const long t0 = g0 * TE_L;                                              // this group's first token row
        const int nt = nb * TE_L;
        // ---- x = relu(LayerNorm(first_layer(tiles)))
        if (SAVE) te_dump<64>(Nb, TE_PX, sv.tiles64, t0, nt, tid);
        te_gemm<64, 64, 0>(Nb, TE_PX, w0, V + TE_V0, X, TE_PX, lane, wave);
        __syncthreads();
        if (SAVE) { te_dump<64>(X, TE_PX, sv.a0, t0, nt, tid); __syncthreads(); }   // (the LayerNorm below is in place)
        te_layer_norm<TE_D, true>(X, TE_PX, X, TE_PX, V + TE_V0 + 64, V + TE_V0 + 128, tid);
        __syncthreads();
#pragma unroll 1
        for (int l = 0; l < 2; l++) {
            const unsigned short* wl = wts + TE_WL + l * TE_WL_SIZE;
            const float* vl = V + TE_VL + l * TE_VL_SIZE;
            // x = x + out_proj(attention(qkv(LayerNorm(x))))
            TeW<64, 192> wq; te_fetch<64, 192>(wq, wl, lane, wave);
            if (SAVE) te_dump<64>(X, TE_PX, sv.xin[l], t0, nt, tid);
            te_layer_norm<TE_D, false>(X, TE_PX, Nb, TE_PX, vl, vl + 64, tid);
            __syncthreads();
            if (SAVE && sv.n1[l] != nullptr) te_dump<64>(Nb, TE_PX, sv.n1[l], t0, nt, tid);   // (optional: the backward can recompute it)
            te_gemm<64, 192, 0>(Nb, TE_PX, wq, vl + 128, Q, TE_PQ, lane, wave);
            TeW<64, 64> wo; te_fetch<64, 64>(wo, wl + 192 * 64, lane, wave);
            __syncthreads();
            if (SAVE) te_dump<192>(Q, TE_PQ, sv.qkv[l], t0, nt, tid);
            te_attention(Q, Nb, lane, wave);
            __syncthreads();
            if (SAVE) te_dump<64>(Nb, TE_PX, sv.o[l], t0, nt, tid);
            te_gemm<64, 64, 2>(Nb, TE_PX, wo, vl + 320, X, TE_PX, lane, wave);
            TeW<64, 128> w1; te_fetch<64, 128>(w1, wl + 192 * 64 + 64 * 64, lane, wave);
            TeW<128, 64> w2; te_fetch<128, 64>(w2, wl + 192 * 64 + 64 * 64 + 128 * 64, lane, wave);
            __syncthreads();
            // x = x + linear2(relu(linear1(LayerNorm(x))))
            if (SAVE) te_dump<64>(X, TE_PX, sv.xmid[l], t0, nt, tid);
            te_layer_norm<TE_D, false>(X, TE_PX, Nb, TE_PX, vl + 384, vl + 448, tid);
            __syncthreads();
            if (SAVE && sv.n2[l] != nullptr) te_dump<64>(Nb, TE_PX, sv.n2[l], t0, nt, tid);
            te_gemm<64, 128, 1>(Nb, TE_PX, w1, vl + 512, Q, TE_PH, lane, wave);
            __syncthreads();
            if (SAVE && sv.h[l] != nullptr) te_dump<128>(Q, TE_PH, sv.h[l], t0, nt, tid);   // (optional: catan_ffn_outproj_bwd_rh recomputes it)
            te_gemm<128, 64, 2>(Q, TE_PH, w2, vl + 640, X, TE_PX, lane, wave);
            __syncthreads();
        }
        // ---- out = relu(LayerNorm25(out_proj(x)))
        {
            TeW<64, 32> wp; te_fetch<64, 32>(wp, wts + TE_WP, lane, wave);
            if (SAVE) te_dump<64>(X, TE_PX, sv.xfin, t0, nt, tid);
            te_gemm<64, 32, 0>(X, TE_PX, wp, V + TE_VP, Q, TE_PX, lane, wave);
        }
        __syncthreads();
        if (SAVE) {
            for (int c = tid; c < nt * TE_OUT; c += TE_THREADS) {
                const int t = c / TE_OUT, i = c - t * TE_OUT;
                sv.p[(t0 + t) * TE_OUT + i] = Q[t * TE_PX + i];
            }
        }
        // (out_pitch elements per board, 19 x 25 of them written: the training path pads the board rows to whole 16-byte pieces)
        te_final_norm(Q, TE_PX, V + TE_VP + 32, V + TE_VP + 64, out + g0 * out_pitch, out_pitch, nb * TE_L, tid);
    }
}

}  // namespace catan
