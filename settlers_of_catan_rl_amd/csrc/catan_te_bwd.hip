// catan_te_bwd.hip - backward of the tile encoder's pointwise sub-layer  x_out = x + linear2(relu(linear1(LayerNorm(x))))
// (reference RL/models/tile_encoder.py:41-91 via its transformer layers) for everything but the weight gradients, in ONE pass
// over the token rows.  As separate kernels - (dX W2) masked by the ReLU, (dH W1), LayerNorm backward + residual - the chain
// moved 1 152 bf16 elements per token through HBM (dH written and read twice, dN written and read); here a wave takes 16 tokens
// through the whole chain: reads dX [64], the ReLU output H [128] and the LayerNorm input X [64], writes dH [128] (the two
// weight-gradient kernels read it) and the gradient of X [64].
//   dH^T = W2^T . dX^T      transposed product (weights as the A operand): a lane ends up with dH[token l % 16][16 j + 4 (l / 16) + i],
//                           which - masked by H > 0 and rounded to bf16 - IS the A operand of the next product once its contraction
//                           index is permuted (as k_head_fwd chains its two products)
//   dN  = dH . W1           B fragments = two 8-byte loads of W1^T in the same permuted order
//   dX' = LayerNorm'(dN) + dX   per-row statistics by xor-shuffles over the 16 lanes that hold a row; LayerNorm weight / bias gradients
//                           accumulate in registers over the wave's tiles and leave as one atomic per column and workgroup
// Rounding follows the unfused chain: dH, dN and the LayerNorm part of dX' are rounded to bf16 where the separate kernels stored them.
#pragma once

namespace catan {

constexpr int FB_P = 72;          // LDS row pitch of the 16 x 64 tiles and of W2^T (bf16 elements): 144 B
constexpr int FB_P1 = 136;        // ... of W1^T: 272 B

__global__ __launch_bounds__(256) void k_ffn_bwd_dx(const unsigned short* __restrict__ dx, const unsigned short* __restrict__ h, const unsigned short* __restrict__ x,
                                                    const unsigned short* __restrict__ w2t, const unsigned short* __restrict__ w1t, const float* __restrict__ lnw,
                                                    float eps, unsigned short* __restrict__ dh, unsigned short* __restrict__ dxo, float* __restrict__ dlnw,
                                                    float* __restrict__ dlnb, long rows) {
    __shared__ __attribute__((aligned(16))) unsigned short sD[4][16 * FB_P];     // the tile's dX rows
    __shared__ __attribute__((aligned(16))) unsigned short sX[4][16 * FB_P];     // the tile's X rows; then the outgoing dX' rows
    __shared__ float sG[2][64];
    __shared__ __attribute__((aligned(16))) unsigned short sW2[128 * FB_P];
    __shared__ __attribute__((aligned(16))) unsigned short sW1[64 * FB_P1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, g = lane >> 4;
    if (tid < 128) (&sG[0][0])[tid] = 0.f;
    // weights in LDS (W2^T [128][64], W1^T [64][128]: 36 KB): as register fragments they cost 128 VGPRs and left one wave per SIMD,
    // every global access of a tile fully exposed; from LDS a fragment is one 16-byte (two 8-byte) read per use
    for (int c = tid; c < 128 * 8; c += 256) {
        const int n = c >> 3, ch = c & 7;
        *reinterpret_cast<uint4*>(sW2 + n * FB_P + ch * 8) = *reinterpret_cast<const uint4*>(w2t + n * 64 + ch * 8);
    }
    for (int c = tid; c < 64 * 16; c += 256) {
        const int n = c >> 4, ch = c & 15;
        *reinterpret_cast<uint4*>(sW1 + n * FB_P1 + ch * 8) = *reinterpret_cast<const uint4*>(w1t + n * 128 + ch * 8);
    }
    __syncthreads();
    float wl[4], aw[4] = { 0.f, 0.f, 0.f, 0.f }, ab[4] = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int t = 0; t < 4; t++) wl[t] = lnw[16 * t + lr];
    unsigned short* tD = sD[wave]; unsigned short* tX = sX[wave];
    const long tiles = (rows + 15) / 16;
    // the NEXT tile's rows are requested before the current tile is computed (register double buffer)
    uint4 vd[2], vx[2];
    uint2 hn[8];
    auto request = [&](long tile) {
        const long r0 = tile * 16;
        const long row = r0 + lr < rows ? r0 + lr : rows - 1;                  // rows past the end repeat the last one; nothing is stored for them
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int c = lane + 64 * q, rr = c >> 3, ch = c & 7;
            const long gr = r0 + rr < rows ? r0 + rr : rows - 1;
            vd[q] = *reinterpret_cast<const uint4*>(dx + gr * 64 + ch * 8);
            vx[q] = *reinterpret_cast<const uint4*>(x + gr * 64 + ch * 8);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) hn[j] = *reinterpret_cast<const uint2*>(h + row * 128 + 16 * j + 4 * g);     // the ReLU outputs at this lane's dH positions
    };
    const long step = (long)gridDim.x * 4;
    long tile = (long)blockIdx.x * 4 + wave;
    if (tile < tiles) request(tile);
    for (; tile < tiles; tile += step) {
        const long r0 = tile * 16;
        const long row = r0 + lr < rows ? r0 + lr : rows - 1;
        uint2 hv[8];
#pragma unroll
        for (int j = 0; j < 8; j++) hv[j] = hn[j];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int c = lane + 64 * q, rr = c >> 3, ch = c & 7;
            *reinterpret_cast<uint4*>(tD + rr * FB_P + ch * 8) = vd[q];
            *reinterpret_cast<uint4*>(tX + rr * FB_P + ch * 8) = vx[q];
        }
        if (tile + step < tiles) request(tile + step);
        __builtin_amdgcn_wave_barrier();
        // ---- dH^T = W2^T . dX^T, masked by H > 0
        bf16x8_t db[2];
#pragma unroll
        for (int s = 0; s < 2; s++) db[s] = *reinterpret_cast<const bf16x8_t*>(tD + lr * FB_P + 32 * s + 8 * g);
        unsigned hp[4][4];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            f32x4_t c = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int s = 0; s < 2; s++)
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(sW2 + (16 * j + lr) * FB_P + 32 * s + 8 * g), db[s], c, 0, 0, 0);
            const unsigned h01 = hv[j].x, h23 = hv[j].y;
            // (H is a ReLU output: > 0 <=> its bf16 bits are neither +0 nor negative; the unfused kernel tests the value)
            const unsigned p0 = pk_bf(__uint_as_float(h01 << 16) > 0.f ? c[0] : 0.f, __uint_as_float(h01 & 0xFFFF0000u) > 0.f ? c[1] : 0.f);
            const unsigned p1 = pk_bf(__uint_as_float(h23 << 16) > 0.f ? c[2] : 0.f, __uint_as_float(h23 & 0xFFFF0000u) > 0.f ? c[3] : 0.f);
            hp[j >> 1][(j & 1) * 2] = p0; hp[j >> 1][(j & 1) * 2 + 1] = p1;
            if (r0 + lr < rows) *reinterpret_cast<uint2*>(dh + row * 128 + 16 * j + 4 * g) = make_uint2(p0, p1);
        }
        // ---- dN = dH . W1: lane holds rows 4 g + r, column 16 t + lr
        float dn[4][4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            f32x4_t c = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int s = 0; s < 4; s++) {
                uint4 au; au.x = hp[s][0]; au.y = hp[s][1]; au.z = hp[s][2]; au.w = hp[s][3];
                const unsigned short* wr = sW1 + (16 * t + lr) * FB_P1 + 32 * s + 4 * g;
                const uint2 lo = *reinterpret_cast<const uint2*>(wr), hi = *reinterpret_cast<const uint2*>(wr + 16);
                uint4 bu; bu.x = lo.x; bu.y = lo.y; bu.z = hi.x; bu.w = hi.y;
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(&au), *reinterpret_cast<const bf16x8_t*>(&bu), c, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) dn[t][r] = hd_bf(c[r]);
        }
        // ---- LayerNorm backward over the rows 4 g + r (their 64 columns sit in the 16 lanes of the quarter wave x 4 tiles) + the residual dX
        float xv[4][4], res[4][4];
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                xv[t][r] = te_bf(tX[(4 * g + r) * FB_P + 16 * t + lr]);
                res[t][r] = te_bf(tD[(4 * g + r) * FB_P + 16 * t + lr]);
            }
        __builtin_amdgcn_wave_barrier();                                       // (tX is overwritten with the result below)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float sum = xv[0][r] + xv[1][r] + xv[2][r] + xv[3][r];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) sum += __shfl_xor(sum, m);
            const float mean = sum * (1.f / 64.f);
            float sq = 0.f;
#pragma unroll
            for (int t = 0; t < 4; t++) { xv[t][r] -= mean; sq += xv[t][r] * xv[t][r]; }
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) sq += __shfl_xor(sq, m);
            const float rstd = rsqrtf(sq * (1.f / 64.f) + eps);
            const bool live = r0 + 4 * g + r < rows;
            float gw[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                xv[t][r] *= rstd;                                              // x_hat
                const float gy = live ? dn[t][r] : 0.f;
                aw[t] += gy * xv[t][r]; ab[t] += gy;
                gw[t] = gy * wl[t];
                s1 += gw[t]; s2 += gw[t] * xv[t][r];
            }
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
            const float m1 = s1 * (1.f / 64.f), m2 = s2 * (1.f / 64.f);
#pragma unroll
            for (int t = 0; t < 4; t++)
                tX[(4 * g + r) * FB_P + 16 * t + lr] = te_to_bf(hd_bf(rstd * (gw[t] - m1 - xv[t][r] * m2)) + res[t][r]);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int c = lane + 64 * q, rr = c >> 3, ch = c & 7;
            if (r0 + rr < rows) *reinterpret_cast<uint4*>(dxo + (r0 + rr) * 64 + ch * 8) = *reinterpret_cast<const uint4*>(tX + rr * FB_P + ch * 8);
        }
        __builtin_amdgcn_wave_barrier();                                       // (the next tile's staging overwrites tD / tX)
    }
    // ---- LayerNorm weight / bias gradients: over the four row groups of the wave, the waves of the workgroup, then one atomic per column
#pragma unroll
    for (int t = 0; t < 4; t++) {
        aw[t] += __shfl_xor(aw[t], 16); aw[t] += __shfl_xor(aw[t], 32);
        ab[t] += __shfl_xor(ab[t], 16); ab[t] += __shfl_xor(ab[t], 32);
    }
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int t = 0; t < 4; t++) { atomicAdd(&sG[0][16 * t + lr], aw[t]); atomicAdd(&sG[1][16 * t + lr], ab[t]); }
    }
    __syncthreads();
    if (tid < 64) { atomicAdd(dlnw + tid, sG[0][tid]); atomicAdd(dlnb + tid, sG[1][tid]); }
}


// The attention sub-layer's input side, x_mid = x + out_proj(attention(qkv(LayerNorm(x)))): the gradient of x from dQKV [rows][192]
// (catan_attention_bwd's output) - dN = dQKV . Wqkv, then LayerNorm backward + the residual gradient d(x_mid) - in one pass.  The
// separate kernels wrote dN and read it back (and the residual); here: dQKV, X and the residual in, dX out.  wt = Wqkv^T [64][192].
constexpr int QB_K = 192, QB_PW = QB_K + 8;       // LDS row pitch of Wqkv^T (bf16 elements): 400 B
__global__ __launch_bounds__(256) void k_qkv_bwd_dx(const unsigned short* __restrict__ dqkv, const unsigned short* __restrict__ x, const unsigned short* __restrict__ dres,
                                                    const unsigned short* __restrict__ wt, const float* __restrict__ lnw, float eps,
                                                    unsigned short* __restrict__ dxo, float* __restrict__ dlnw, float* __restrict__ dlnb, long rows) {
    __shared__ __attribute__((aligned(16))) unsigned short sD[4][16 * FB_P];     // the tile's residual-gradient rows
    __shared__ __attribute__((aligned(16))) unsigned short sX[4][16 * FB_P];     // the tile's X rows; then the outgoing dX rows
    __shared__ __attribute__((aligned(16))) unsigned short sW[64 * QB_PW];
    __shared__ float sG[2][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, g = lane >> 4;
    if (tid < 128) (&sG[0][0])[tid] = 0.f;
    for (int c = tid; c < 64 * (QB_K / 8); c += 256) {
        const int n = c / (QB_K / 8), ch = c - n * (QB_K / 8);
        *reinterpret_cast<uint4*>(sW + n * QB_PW + ch * 8) = *reinterpret_cast<const uint4*>(wt + n * QB_K + ch * 8);
    }
    __syncthreads();
    float wl[4], aw[4] = { 0.f, 0.f, 0.f, 0.f }, ab[4] = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int t = 0; t < 4; t++) wl[t] = lnw[16 * t + lr];
    unsigned short* tD = sD[wave]; unsigned short* tX = sX[wave];
    const long tiles = (rows + 15) / 16;
    uint4 vd[2], vx[2], an[QB_K / 32];
    auto request = [&](long tile) {
        const long r0 = tile * 16;
        const long row = r0 + lr < rows ? r0 + lr : rows - 1;                  // rows past the end repeat the last one; nothing is stored for them
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int c = lane + 64 * q, rr = c >> 3, ch = c & 7;
            const long gr = r0 + rr < rows ? r0 + rr : rows - 1;
            vd[q] = *reinterpret_cast<const uint4*>(dres + gr * 64 + ch * 8);
            vx[q] = *reinterpret_cast<const uint4*>(x + gr * 64 + ch * 8);
        }
#pragma unroll
        for (int s = 0; s < QB_K / 32; s++) an[s] = *reinterpret_cast<const uint4*>(dqkv + row * QB_K + 32 * s + 8 * g);     // A fragments straight from the rows
    };
    const long step = (long)gridDim.x * 4;
    long tile = (long)blockIdx.x * 4 + wave;
    if (tile < tiles) request(tile);
    for (; tile < tiles; tile += step) {
        const long r0 = tile * 16;
        uint4 a[QB_K / 32];
#pragma unroll
        for (int s = 0; s < QB_K / 32; s++) a[s] = an[s];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int c = lane + 64 * q, rr = c >> 3, ch = c & 7;
            *reinterpret_cast<uint4*>(tD + rr * FB_P + ch * 8) = vd[q];
            *reinterpret_cast<uint4*>(tX + rr * FB_P + ch * 8) = vx[q];
        }
        if (tile + step < tiles) request(tile + step);
        __builtin_amdgcn_wave_barrier();
        // ---- dN = dQKV . Wqkv: lane holds rows 4 g + r, column 16 t + lr
        float dn[4][4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            f32x4_t c = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int s = 0; s < QB_K / 32; s++)
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(&a[s]),
                                                            *reinterpret_cast<const bf16x8_t*>(sW + (16 * t + lr) * QB_PW + 32 * s + 8 * g), c, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) dn[t][r] = hd_bf(c[r]);
        }
        float xv[4][4], res[4][4];
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                xv[t][r] = te_bf(tX[(4 * g + r) * FB_P + 16 * t + lr]);
                res[t][r] = te_bf(tD[(4 * g + r) * FB_P + 16 * t + lr]);
            }
        __builtin_amdgcn_wave_barrier();                                       // (tX is overwritten with the result below)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float sum = xv[0][r] + xv[1][r] + xv[2][r] + xv[3][r];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) sum += __shfl_xor(sum, m);
            const float mean = sum * (1.f / 64.f);
            float sq = 0.f;
#pragma unroll
            for (int t = 0; t < 4; t++) { xv[t][r] -= mean; sq += xv[t][r] * xv[t][r]; }
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) sq += __shfl_xor(sq, m);
            const float rstd = rsqrtf(sq * (1.f / 64.f) + eps);
            const bool live = r0 + 4 * g + r < rows;
            float gw[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                xv[t][r] *= rstd;                                              // x_hat
                const float gy = live ? dn[t][r] : 0.f;
                aw[t] += gy * xv[t][r]; ab[t] += gy;
                gw[t] = gy * wl[t];
                s1 += gw[t]; s2 += gw[t] * xv[t][r];
            }
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
            const float m1 = s1 * (1.f / 64.f), m2 = s2 * (1.f / 64.f);
#pragma unroll
            for (int t = 0; t < 4; t++)
                tX[(4 * g + r) * FB_P + 16 * t + lr] = te_to_bf(hd_bf(rstd * (gw[t] - m1 - xv[t][r] * m2)) + res[t][r]);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int c = lane + 64 * q, rr = c >> 3, ch = c & 7;
            if (r0 + rr < rows) *reinterpret_cast<uint4*>(dxo + (r0 + rr) * 64 + ch * 8) = *reinterpret_cast<const uint4*>(tX + rr * FB_P + ch * 8);
        }
        __builtin_amdgcn_wave_barrier();                                       // (the next tile's staging overwrites tD / tX)
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        aw[t] += __shfl_xor(aw[t], 16); aw[t] += __shfl_xor(aw[t], 32);
        ab[t] += __shfl_xor(ab[t], 16); ab[t] += __shfl_xor(ab[t], 32);
    }
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int t = 0; t < 4; t++) { atomicAdd(&sG[0][16 * t + lr], aw[t]); atomicAdd(&sG[1][16 * t + lr], ab[t]); }
    }
    __syncthreads();
    if (tid < 64) { atomicAdd(dlnw + tid, sG[0][tid]); atomicAdd(dlnb + tid, sG[1][tid]); }
}


// k_ffn_bwd_w: k_ffn_bwd_dx AND the sub-layer's two weight gradients in one pass over the rows.  The weight gradients
// (catan_linear_wgrad: dW2 = dX^T H, dW1 = dH^T N) re-read dX, H, dH and N from HBM although k_ffn_bwd_dx has three of them on chip:
// 384 of the 832 bf16 elements the chain moved per token.  Here a workgroup takes 64 token rows per stage: dX, H, X and N are staged
// in LDS in k_wgrad_tr's sub-tile image (row-major inside 32 x 16 sub-tiles: a row fragment is one 16-byte read, a transposed
// fragment one ds_read_tr16_b64 pair), every wave takes its 16 rows through k_ffn_bwd_dx's chain - dH goes into an LDS image instead
// of HBM - and then, over all 64 rows, accumulates its quarter of dW2 (one 16-row tile of the 64 outputs x 128 + 1 columns; the
// extra column of ones yields db2) and of dW1 (two tiles of the 128 x 64 + 1) on MFMA; the accumulators leave by fp32 atomics at
// the end, as in k_wgrad_tr.  HBM: dX, H, X, N in, dX' out: 384 elements per token.
constexpr int FW_ROWS = 64;
constexpr int FW_IMG64 = 4 * 2 * WG_SUB, FW_IMG65 = 5 * 2 * WG_SUB, FW_IMG128 = 8 * 2 * WG_SUB, FW_IMG129 = 9 * 2 * WG_SUB;   // elements per image
DEVI int fw_off(int row, int col) { return wg_sub_off(col >> 4, row >> 5) + (row & 31) * 16 + (col & 15); }
// LayerNorm backward + residual of one token per lane (k_ffn_bwd_w).  The dN product is formed TRANSPOSED
// (weights as the A operand), so lane (lr, g) holds dN[token lr][16 t + 4 g + i]: a token's 64 columns sit in the four lanes lr,
// lr + 16, lr + 32, lr + 48 as 4 x 4 CONSECUTIVE columns each - X, the residual gradient and the outgoing dX' (and N, when it is
// recomputed) move as 8-byte LDS accesses and the row statistics are two xor-shuffles (the row-per-four-lanes layout of
// k_ffn_bwd_dx takes sixteen 2-byte accesses per image and four shuffles per statistic for each of its four rows): k_ffn_bwd_w<true>
// 1.56 -> 1.37 ms at 3.9 M rows, and recomputing N costs nothing any more (1.35 ms).  k_qkv_bwd_w keeps the row layout: with 64
// instead of 16 registers of LayerNorm weights / gradient sums it needs 286 VGPRs and loses its second workgroup per CU (0.77 -> 0.84 ms).
// xs: the X image (its rows are replaced by dX'); rs: the residual-gradient image; ns: the N image (RN) or nullptr.
template <bool RN>
DEVI void fw_ln_bwd_token(unsigned short* xs, const unsigned short* rs, unsigned short* ns, int row, int g, bool live, const float (&dn)[4][4],
                          const float (&wl)[4][4], const float (&bl)[4][4], float eps, float (&aw)[4][4], float (&ab)[4][4]) {
    float xv[4][4], res[4][4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int o = fw_off(row, 16 * t + 4 * g);
        const uint2 ux = *reinterpret_cast<const uint2*>(xs + o), ur = *reinterpret_cast<const uint2*>(rs + o);
        xv[t][0] = __uint_as_float(ux.x << 16); xv[t][1] = __uint_as_float(ux.x & 0xFFFF0000u);
        xv[t][2] = __uint_as_float(ux.y << 16); xv[t][3] = __uint_as_float(ux.y & 0xFFFF0000u);
        res[t][0] = __uint_as_float(ur.x << 16); res[t][1] = __uint_as_float(ur.x & 0xFFFF0000u);
        res[t][2] = __uint_as_float(ur.y << 16); res[t][3] = __uint_as_float(ur.y & 0xFFFF0000u);
    }
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int i = 0; i < 4; i++) sum += xv[t][i];
    sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.f / 64.f);
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int i = 0; i < 4; i++) { xv[t][i] -= mean; sq += xv[t][i] * xv[t][i]; }
    sq += __shfl_xor(sq, 16); sq += __shfl_xor(sq, 32);
    const float rstd = rsqrtf(sq * (1.f / 64.f) + eps);
    float gw[4][4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        unsigned short nb[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            xv[t][i] *= rstd;
            if (RN) nb[i] = live ? te_to_bf(xv[t][i] * wl[t][i] + bl[t][i]) : (unsigned short)0;
            const float gy = live ? dn[t][i] : 0.f;
            aw[t][i] += gy * xv[t][i]; ab[t][i] += gy;
            gw[t][i] = gy * wl[t][i];
            s1 += gw[t][i]; s2 += gw[t][i] * xv[t][i];
        }
        if (RN) *reinterpret_cast<uint2*>(ns + fw_off(row, 16 * t + 4 * g)) = make_uint2((unsigned)nb[0] | ((unsigned)nb[1] << 16), (unsigned)nb[2] | ((unsigned)nb[3] << 16));
    }
    s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
    const float m1 = s1 * (1.f / 64.f), m2 = s2 * (1.f / 64.f);
#pragma unroll
    for (int t = 0; t < 4; t++) {
        unsigned short ob[4];
#pragma unroll
        for (int i = 0; i < 4; i++) ob[i] = te_to_bf(hd_bf(rstd * (gw[t][i] - m1 - xv[t][i] * m2)) + res[t][i]);
        *reinterpret_cast<uint2*>(xs + fw_off(row, 16 * t + 4 * g)) = make_uint2((unsigned)ob[0] | ((unsigned)ob[1] << 16), (unsigned)ob[2] | ((unsigned)ob[3] << 16));
    }
}
// LayerNorm FORWARD of one token per lane in the same layout (RH: the FFN's input N is needed before the chain starts, to recompute
// H = relu(N W1^T + b1)): N = ((x - mean) rstd) w + b rounded to bf16 as the forward kernel's te_layer_norm stores it; a dead row gives 0.
DEVI void fw_ln_fwd_token(const unsigned short* xs, unsigned short* ns, int row, int g, bool live, const float (&wl)[4][4], const float (&bl)[4][4], float eps) {
    float xv[4][4], sum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const uint2 ux = *reinterpret_cast<const uint2*>(xs + fw_off(row, 16 * t + 4 * g));
        xv[t][0] = __uint_as_float(ux.x << 16); xv[t][1] = __uint_as_float(ux.x & 0xFFFF0000u);
        xv[t][2] = __uint_as_float(ux.y << 16); xv[t][3] = __uint_as_float(ux.y & 0xFFFF0000u);
#pragma unroll
        for (int i = 0; i < 4; i++) sum += xv[t][i];
    }
    sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.f / 64.f);
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int i = 0; i < 4; i++) { xv[t][i] -= mean; sq += xv[t][i] * xv[t][i]; }
    sq += __shfl_xor(sq, 16); sq += __shfl_xor(sq, 32);
    const float rstd = rsqrtf(sq * (1.f / 64.f) + eps);
#pragma unroll
    for (int t = 0; t < 4; t++) {
        float y[4];
#pragma unroll
        for (int i = 0; i < 4; i++) y[i] = live ? xv[t][i] * rstd * wl[t][i] + bl[t][i] : 0.f;
        *reinterpret_cast<uint2*>(ns + fw_off(row, 16 * t + 4 * g)) = make_uint2(pk_bf(y[0], y[1]), pk_bf(y[2], y[3]));
    }
}
// the LayerNorm weight / bias gradients of that layout: summed over the 16 token lanes, one LDS atomic per column and wave
DEVI void fw_ln_grads_out(float (&aw)[4][4], float (&ab)[4][4], float (*sG)[64], int lr, int g) {
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { aw[t][i] += __shfl_xor(aw[t][i], m); ab[t][i] += __shfl_xor(ab[t][i], m); }
            if (lr == 0) { atomicAdd(&sG[0][16 * t + 4 * g + i], aw[t][i]); atomicAdd(&sG[1][16 * t + 4 * g + i], ab[t][i]); }
        }
}
// OP: also the backward of the out-projection that FEEDS this sub-layer's input (x = x_in + o Wo^T + bo: the gradient of x is the
// gradient of that product's output): dO = dX' Wo and dWo = dX'^T O from the dX' rows while they are in LDS - the separate row
// product and weight-gradient kernels read dX' twice more.  o [rows][64] = the attention output, wot = Wo^T [64 in][64 out],
// d_o [rows][64] out, dwo [64][64] / dbo [64] accumulated.
struct FfnOutProj { const unsigned short* o; const unsigned short* wot; unsigned short* d_o; float* dwo; float* dbo; };
// RN: N is not read but recomputed from X (LayerNorm weight lnw, bias lnb: the forward's formula on the statistics the LayerNorm
// backward forms anyway), so the training forward need not store it: 128 of the 896 bytes per token row this pass read, and 128 of
// the 1 280 the forward wrote per row and layer.
// RH: H is not read either but recomputed - H = relu(N W1^T + b1) from the recomputed N, one more 16 x 64 x 128 product per wave and
// stage (w1 = W1 [128][64] row-major as the forward reads it, b1 [128] as the forward adds it) - so the training forward need not
// store its widest activation: 256 of the 1 024 bytes per token and layer it wrote, and 256 of the 768 this pass read.
struct FfnRecomputeH { const unsigned short* w1; const float* b1; };
template <bool OP, bool RN, bool RH = false>
__global__ __launch_bounds__(256) void k_ffn_bwd_w(const unsigned short* __restrict__ dx, const unsigned short* __restrict__ h, const unsigned short* __restrict__ x,
                                                   const unsigned short* __restrict__ n2, const unsigned short* __restrict__ w2t, const unsigned short* __restrict__ w1t,
                                                   const float* __restrict__ lnw, const float* __restrict__ lnb, float eps, unsigned short* __restrict__ dxo,
                                                   float* __restrict__ dw2, float* __restrict__ db2, float* __restrict__ dw1, float* __restrict__ db1,
                                                   float* __restrict__ dlnw, float* __restrict__ dlnb, long rows, long rows_per_block, FfnOutProj op,
                                                   FfnRecomputeH rh = FfnRecomputeH{nullptr, nullptr}) {
    static_assert(!RH || RN, "H is recomputed from the recomputed N");
    __shared__ __attribute__((aligned(16))) unsigned short sW1n[RH ? 128 * FB_P : 8];   // W1 [128][64] (RH)
    __shared__ __attribute__((aligned(16))) float sB1[RH ? 128 : 4];
    __shared__ __attribute__((aligned(16))) unsigned short sDX[FW_IMG64];      // dX; the rows of dX' replace X below
    __shared__ __attribute__((aligned(16))) unsigned short sX[FW_IMG64];
    __shared__ __attribute__((aligned(16))) unsigned short sH[FW_IMG129];      // H and the column of ones
    __shared__ __attribute__((aligned(16))) unsigned short sN[FW_IMG65];       // N = LayerNorm(X) and the column of ones
    __shared__ __attribute__((aligned(16))) unsigned short sDH[FW_IMG128];
    __shared__ __attribute__((aligned(16))) unsigned short sW2[128 * FB_P];
    __shared__ __attribute__((aligned(16))) unsigned short sW1[64 * FB_P1];
    __shared__ __attribute__((aligned(16))) unsigned short sO[OP ? FW_IMG65 : 8];        // O and the column of ones
    __shared__ __attribute__((aligned(16))) unsigned short sWo[OP ? 64 * FB_P : 8];
    __shared__ float sG[2][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, g = lane >> 4;
    const long r_begin = (long)blockIdx.x * rows_per_block;
    const long r_end = r_begin + rows_per_block < rows ? r_begin + rows_per_block : rows;
    if (r_begin >= rows) return;
    if (tid < 128) (&sG[0][0])[tid] = 0.f;
    if (OP) {
        for (int c = tid; c < 64 * 8; c += 256) { const int n = c >> 3, ch = c & 7; *reinterpret_cast<uint4*>(sWo + n * FB_P + ch * 8) = *reinterpret_cast<const uint4*>(op.wot + n * 64 + ch * 8); }
        for (int c = tid; c < FW_IMG65; c += 256) sO[c] = 0;
    }
    for (int c = tid; c < 128 * 8; c += 256) { const int n = c >> 3, ch = c & 7; *reinterpret_cast<uint4*>(sW2 + n * FB_P + ch * 8) = *reinterpret_cast<const uint4*>(w2t + n * 64 + ch * 8); }
    for (int c = tid; c < 64 * 16; c += 256) { const int n = c >> 4, ch = c & 15; *reinterpret_cast<uint4*>(sW1 + n * FB_P1 + ch * 8) = *reinterpret_cast<const uint4*>(w1t + n * 128 + ch * 8); }
    if (RH) {
        for (int c = tid; c < 128 * 8; c += 256) { const int n = c >> 3, ch = c & 7; *reinterpret_cast<uint4*>(sW1n + n * FB_P + ch * 8) = *reinterpret_cast<const uint4*>(rh.w1 + n * 64 + ch * 8); }
        if (tid < 128) sB1[tid] = rh.b1[tid];
    }
    for (int c = tid; c < FW_IMG129; c += 256) sH[c] = 0;                        // (the ones columns' tiles: everything but column 0 stays zero)
    for (int c = tid; c < FW_IMG65; c += 256) sN[c] = 0;
    float wl[4][4], bl[4][4], aw[4][4], ab[4][4];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int i = 0; i < 4; i++) { wl[t][i] = lnw[16 * t + 4 * g + i]; bl[t][i] = RN ? lnb[16 * t + 4 * g + i] : 0.f; aw[t][i] = 0.f; ab[t][i] = 0.f; }
    f32x4_t acc2[9], acc1[2][5];
#pragma unroll
    for (int b = 0; b < 9; b++) acc2[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 5; b++) acc1[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    f32x4_t acco[5];
#pragma unroll
    for (int b = 0; b < 5; b++) acco[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    uint4 vdx[2], vx[2], vn[2], vh[4], vo[2];
    if (OP) wg_load<2>(op.o, r_begin * 64, rows * 64, 64, vo, tid);
    wg_load<2>(dx, r_begin * 64, rows * 64, 64, vdx, tid);
    wg_load<2>(x, r_begin * 64, rows * 64, 64, vx, tid);
    if (!RN) wg_load<2>(n2, r_begin * 64, rows * 64, 64, vn, tid);
    if (!RH) wg_load<4>(h, r_begin * 128, rows * 128, 128, vh, tid);
    __syncthreads();
    const int row = 16 * wave + lr;                                             // this lane's token row of the stage (operand layout)
    for (long r0 = r_begin; r0 < r_end; r0 += FW_ROWS) {
        wg_store_rows<2>(sDX, 64, vdx, tid);
        wg_store_rows<2>(sX, 64, vx, tid);
        if (!RN) wg_store_rows<2>(sN, 64, vn, tid);
        if (!RH) wg_store_rows<4>(sH, 128, vh, tid);
        if (OP) wg_store_rows<2>(sO, 64, vo, tid);
        if (tid < FW_ROWS) {
            const unsigned short one = (r0 + tid < r_end) ? (unsigned short)0x3F80 : (unsigned short)0;   // bf16 1.0: the bias columns
            sH[fw_off(tid, 128)] = one; sN[fw_off(tid, 64)] = one;
            if (OP) sO[fw_off(tid, 64)] = one;
        }
        __syncthreads();
        if (r0 + FW_ROWS < r_end) {                                              // the next stage's rows fly during this stage
            if (OP) wg_load<2>(op.o, (r0 + FW_ROWS) * 64, rows * 64, 64, vo, tid);
            wg_load<2>(dx, (r0 + FW_ROWS) * 64, rows * 64, 64, vdx, tid);
            wg_load<2>(x, (r0 + FW_ROWS) * 64, rows * 64, 64, vx, tid);
            if (!RN) wg_load<2>(n2, (r0 + FW_ROWS) * 64, rows * 64, 64, vn, tid);
            if (!RH) wg_load<4>(h, (r0 + FW_ROWS) * 128, rows * 128, 128, vh, tid);
        }
        uint2 hreg[8];
        if (RH) {
            // ---- N = LayerNorm(X) for this wave's 16 rows, then H^T = W1 . N^T + b1, ReLU: transposed product, so the lane ends up with
            //      H[token `row`][16 j + 4 g + i] - the positions whose dH it forms below - and writes them into the H image for dW2
            fw_ln_fwd_token(sX, sN, row, g, r0 + row < r_end, wl, bl, eps);
            __builtin_amdgcn_wave_barrier();
            bf16x8_t nb[2];
#pragma unroll
            for (int s = 0; s < 2; s++) nb[s] = *reinterpret_cast<const bf16x8_t*>(sN + fw_off(row, 32 * s + 8 * g));
            const bool live_row = r0 + row < r_end;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float4 bv = *reinterpret_cast<const float4*>(sB1 + 16 * j + 4 * g);
                f32x4_t c = { bv.x, bv.y, bv.z, bv.w };
#pragma unroll
                for (int s = 0; s < 2; s++)
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(sW1n + (16 * j + lr) * FB_P + 32 * s + 8 * g), nb[s], c, 0, 0, 0);
                hreg[j] = live_row ? make_uint2(pk_bf(fmaxf(c[0], 0.f), fmaxf(c[1], 0.f)), pk_bf(fmaxf(c[2], 0.f), fmaxf(c[3], 0.f))) : make_uint2(0u, 0u);
                *reinterpret_cast<uint2*>(sH + fw_off(row, 16 * j + 4 * g)) = hreg[j];
            }
        }
        // ---- dH^T = W2^T . dX^T, masked by H > 0 (this wave's 16 rows), into the dH image
        bf16x8_t db[2];
#pragma unroll
        for (int s = 0; s < 2; s++) db[s] = *reinterpret_cast<const bf16x8_t*>(sDX + fw_off(row, 32 * s + 8 * g));
        unsigned hp[4][4];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            f32x4_t c = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int s = 0; s < 2; s++)
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(sW2 + (16 * j + lr) * FB_P + 32 * s + 8 * g), db[s], c, 0, 0, 0);
            const uint2 hv = RH ? hreg[j] : *reinterpret_cast<const uint2*>(sH + fw_off(row, 16 * j + 4 * g));
            const unsigned p0 = pk_bf(__uint_as_float(hv.x << 16) > 0.f ? c[0] : 0.f, __uint_as_float(hv.x & 0xFFFF0000u) > 0.f ? c[1] : 0.f);
            const unsigned p1 = pk_bf(__uint_as_float(hv.y << 16) > 0.f ? c[2] : 0.f, __uint_as_float(hv.y & 0xFFFF0000u) > 0.f ? c[3] : 0.f);
            hp[j >> 1][(j & 1) * 2] = p0; hp[j >> 1][(j & 1) * 2 + 1] = p1;
            *reinterpret_cast<uint2*>(sDH + fw_off(row, 16 * j + 4 * g)) = make_uint2(p0, p1);
        }
        // ---- dN^T = W1^T . dH^T (the weights as the A operand again): lane holds token `row`, columns 16 t + 4 g + i
        float dn[4][4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            f32x4_t c = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int s = 0; s < 4; s++) {
                uint4 au; au.x = hp[s][0]; au.y = hp[s][1]; au.z = hp[s][2]; au.w = hp[s][3];
                const unsigned short* wr = sW1 + (16 * t + lr) * FB_P1 + 32 * s + 4 * g;
                const uint2 lo = *reinterpret_cast<const uint2*>(wr), hi = *reinterpret_cast<const uint2*>(wr + 16);
                uint4 bu; bu.x = lo.x; bu.y = lo.y; bu.z = hi.x; bu.w = hi.y;
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(&bu), *reinterpret_cast<const bf16x8_t*>(&au), c, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) dn[t][i] = hd_bf(c[i]);
        }
        // ---- LayerNorm backward + the residual dX; the result replaces the wave's rows of the X image
        fw_ln_bwd_token<RN && !RH>(sX, sDX, (RN && !RH) ? sN : nullptr, row, g, r0 + row < r_end, dn, wl, bl, eps, aw, ab);   // (RH: N is in its image already)
        if (OP) {
            // ---- dO = dX' Wo for this wave's 16 rows, from the rows of dX' the wave has just written (its stores fly during the weight gradients)
            __builtin_amdgcn_wave_barrier();
            bf16x8_t ax[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) ax[s2] = *reinterpret_cast<const bf16x8_t*>(sX + fw_off(row, 32 * s2 + 8 * g));
            f32x4_t co[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                co[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++)
                    co[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(sWo + (16 * t + lr) * FB_P + 32 * s2 + 8 * g), ax[s2], co[t], 0, 0, 0);
            }
            // (transposed product: the lane holds token `row`, columns 16 t + 4 g + i - 8-byte stores straight to HBM, a token's 128 bytes
            //  from its four lanes; no image, no barrier)
            if (r0 + row < r_end) {
#pragma unroll
                for (int t = 0; t < 4; t++)
                    *reinterpret_cast<uint2*>(op.d_o + (r0 + row) * 64 + 16 * t + 4 * g) = make_uint2(pk_bf(co[t][0], co[t][1]), pk_bf(co[t][2], co[t][3]));
            }
        }
        __syncthreads();                                                         // dH image and the dX' rows complete
        // ---- the stage's rows of dX' leave (16-byte pieces of the image rows)
        for (int c = tid; c < FW_ROWS * 8; c += 256) {
            const int rr = c >> 3, ch = c & 7;
            if (r0 + rr < r_end) *reinterpret_cast<uint4*>(dxo + (r0 + rr) * 64 + ch * 8) = *reinterpret_cast<const uint4*>(sX + fw_off(rr, ch * 8));
        }
        // ---- weight gradients over the stage's 64 rows: dW2 tile `wave` (16 outputs) x 9 column tiles, dW1 tiles 2 wave, 2 wave + 1 x 5
#pragma unroll
        for (int ks = 0; ks < FW_ROWS / 32; ks++) {
            const bf16x8_t a2 = wg_frag_tr(sDX, wave, ks, lane);
#pragma unroll
            for (int b = 0; b < 9; b++) acc2[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, wg_frag_tr(sH, b, ks, lane), acc2[b], 0, 0, 0);
            bf16x8_t bn[5];
#pragma unroll
            for (int b = 0; b < 5; b++) bn[b] = wg_frag_tr(sN, b, ks, lane);
#pragma unroll
            for (int a = 0; a < 2; a++) {
                const bf16x8_t a1f = wg_frag_tr(sDH, 2 * wave + a, ks, lane);
#pragma unroll
                for (int b = 0; b < 5; b++) acc1[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1f, bn[b], acc1[a][b], 0, 0, 0);
            }
            if (OP) {                                                            // dWo tile `wave` (16 outputs) x 5 column tiles (O and the ones)
                const bf16x8_t ao = wg_frag_tr(sX, wave, ks, lane);
#pragma unroll
                for (int b = 0; b < 5; b++) acco[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ao, wg_frag_tr(sO, b, ks, lane), acco[b], 0, 0, 0);
            }
        }
        __syncthreads();                                                         // (the next stage overwrites the images)
    }
    if (OP) {
#pragma unroll
        for (int b = 0; b < 5; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = wave * 16 + 4 * g + r, i = b * 16 + lr;
                const float v = acco[b][r];
                if (v != 0.0f) { if (i < 64) atomicAdd(&op.dwo[o * 64 + i], v); else if (i == 64) atomicAdd(&op.dbo[o], v); }
            }
    }
    // ---- accumulators -> global (fp32 atomics; zeroed by the caller)
#pragma unroll
    for (int b = 0; b < 9; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int o = wave * 16 + 4 * g + r, i = b * 16 + lr;
            const float v = acc2[b][r];
            if (v != 0.0f) { if (i < 128) atomicAdd(&dw2[o * 128 + i], v); else if (i == 128) atomicAdd(&db2[o], v); }
        }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 5; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = (2 * wave + a) * 16 + 4 * g + r, i = b * 16 + lr;
                const float v = acc1[a][b][r];
                if (v != 0.0f) { if (i < 64) atomicAdd(&dw1[o * 64 + i], v); else if (i == 64) atomicAdd(&db1[o], v); }
            }
    fw_ln_grads_out(aw, ab, sG, lr, g);
    __syncthreads();
    if (tid < 64) { atomicAdd(dlnw + tid, sG[0][tid]); atomicAdd(dlnb + tid, sG[1][tid]); }
}


// k_qkv_bwd_w: k_qkv_bwd_dx AND the QKV product's weight gradient (dWqkv = dQKV^T N, N = LayerNorm 1's output) in one pass, built as
// k_ffn_bwd_w: 64-row stages, dQKV / X / the residual gradient / N in LDS images, wave w accumulates output tiles 3 w .. 3 w + 2 of
// the 192 x (64 + 1) gradient.  HBM: dQKV, X, the residual gradient, N in, dX out (the separate weight-gradient kernel read dQKV and N again).
constexpr int FW_IMG192 = 12 * 2 * WG_SUB;
template <bool RN>                 // RN: N recomputed from X instead of read (see k_ffn_bwd_w)
__global__ __launch_bounds__(256) void k_qkv_bwd_w(const unsigned short* __restrict__ dqkv, const unsigned short* __restrict__ x, const unsigned short* __restrict__ dres,
                                                   const unsigned short* __restrict__ n1, const unsigned short* __restrict__ wt, const float* __restrict__ lnw,
                                                   const float* __restrict__ lnb, float eps,
                                                   unsigned short* __restrict__ dxo, float* __restrict__ dw, float* __restrict__ dbias,
                                                   float* __restrict__ dlnw, float* __restrict__ dlnb, long rows, long rows_per_block) {
    __shared__ __attribute__((aligned(16))) unsigned short sDQ[FW_IMG192];
    __shared__ __attribute__((aligned(16))) unsigned short sX[FW_IMG64];       // X; the rows of dX replace it below
    __shared__ __attribute__((aligned(16))) unsigned short sR[FW_IMG64];       // the residual gradient
    __shared__ __attribute__((aligned(16))) unsigned short sN[FW_IMG65];       // N and the column of ones
    __shared__ __attribute__((aligned(16))) unsigned short sW[64 * QB_PW];
    __shared__ float sG[2][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, g = lane >> 4;
    const long r_begin = (long)blockIdx.x * rows_per_block;
    const long r_end = r_begin + rows_per_block < rows ? r_begin + rows_per_block : rows;
    if (r_begin >= rows) return;
    if (tid < 128) (&sG[0][0])[tid] = 0.f;
    for (int c = tid; c < 64 * (QB_K / 8); c += 256) {
        const int n = c / (QB_K / 8), ch = c - n * (QB_K / 8);
        *reinterpret_cast<uint4*>(sW + n * QB_PW + ch * 8) = *reinterpret_cast<const uint4*>(wt + n * QB_K + ch * 8);
    }
    for (int c = tid; c < FW_IMG65; c += 256) sN[c] = 0;
    float wl[4], bl[4] = { 0.f, 0.f, 0.f, 0.f }, aw[4] = { 0.f, 0.f, 0.f, 0.f }, ab[4] = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int t = 0; t < 4; t++) { wl[t] = lnw[16 * t + lr]; if (RN) bl[t] = lnb[16 * t + lr]; }
    f32x4_t acc[3][5];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 5; b++) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    uint4 vq[6], vx[2], vr[2], vn[2];
    wg_load<6>(dqkv, r_begin * QB_K, rows * QB_K, QB_K, vq, tid);
    wg_load<2>(x, r_begin * 64, rows * 64, 64, vx, tid);
    wg_load<2>(dres, r_begin * 64, rows * 64, 64, vr, tid);
    if (!RN) wg_load<2>(n1, r_begin * 64, rows * 64, 64, vn, tid);
    __syncthreads();
    const int row = 16 * wave + lr;
    for (long r0 = r_begin; r0 < r_end; r0 += FW_ROWS) {
        wg_store_rows<6>(sDQ, QB_K, vq, tid);
        wg_store_rows<2>(sX, 64, vx, tid);
        wg_store_rows<2>(sR, 64, vr, tid);
        if (!RN) wg_store_rows<2>(sN, 64, vn, tid);
        if (tid < FW_ROWS) sN[fw_off(tid, 64)] = (r0 + tid < r_end) ? (unsigned short)0x3F80 : (unsigned short)0;
        __syncthreads();
        if (r0 + FW_ROWS < r_end) {
            wg_load<6>(dqkv, (r0 + FW_ROWS) * QB_K, rows * QB_K, QB_K, vq, tid);
            wg_load<2>(x, (r0 + FW_ROWS) * 64, rows * 64, 64, vx, tid);
            wg_load<2>(dres, (r0 + FW_ROWS) * 64, rows * 64, 64, vr, tid);
            if (!RN) wg_load<2>(n1, (r0 + FW_ROWS) * 64, rows * 64, 64, vn, tid);
        }
        // ---- dN = dQKV . Wqkv for this wave's 16 rows
        float dn[4][4];
        {
            bf16x8_t a[QB_K / 32];
#pragma unroll
            for (int s = 0; s < QB_K / 32; s++) a[s] = *reinterpret_cast<const bf16x8_t*>(sDQ + fw_off(row, 32 * s + 8 * g));
#pragma unroll
            for (int t = 0; t < 4; t++) {
                f32x4_t c = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
                for (int s = 0; s < QB_K / 32; s++)
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s], *reinterpret_cast<const bf16x8_t*>(sW + (16 * t + lr) * QB_PW + 32 * s + 8 * g), c, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) dn[t][r] = hd_bf(c[r]);
            }
        }
        float xv[4][4], res[4][4];
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = fw_off(16 * wave + 4 * g + r, 16 * t + lr);
                xv[t][r] = te_bf(sX[o]); res[t][r] = te_bf(sR[o]);
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float sum = xv[0][r] + xv[1][r] + xv[2][r] + xv[3][r];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) sum += __shfl_xor(sum, m);
            const float mean = sum * (1.f / 64.f);
            float sq = 0.f;
#pragma unroll
            for (int t = 0; t < 4; t++) { xv[t][r] -= mean; sq += xv[t][r] * xv[t][r]; }
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) sq += __shfl_xor(sq, m);
            const float rstd = rsqrtf(sq * (1.f / 64.f) + eps);
            const bool live = r0 + 16 * wave + 4 * g + r < r_end;
            float gw[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                xv[t][r] *= rstd;
                if (RN) sN[fw_off(16 * wave + 4 * g + r, 16 * t + lr)] = live ? te_to_bf(xv[t][r] * wl[t] + bl[t]) : (unsigned short)0;
                const float gy = live ? dn[t][r] : 0.f;
                aw[t] += gy * xv[t][r]; ab[t] += gy;
                gw[t] = gy * wl[t];
                s1 += gw[t]; s2 += gw[t] * xv[t][r];
            }
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
            const float m1 = s1 * (1.f / 64.f), m2 = s2 * (1.f / 64.f);
#pragma unroll
            for (int t = 0; t < 4; t++)
                sX[fw_off(16 * wave + 4 * g + r, 16 * t + lr)] = te_to_bf(hd_bf(rstd * (gw[t] - m1 - xv[t][r] * m2)) + res[t][r]);
        }
        __syncthreads();
        for (int c = tid; c < FW_ROWS * 8; c += 256) {
            const int rr = c >> 3, ch = c & 7;
            if (r0 + rr < r_end) *reinterpret_cast<uint4*>(dxo + (r0 + rr) * 64 + ch * 8) = *reinterpret_cast<const uint4*>(sX + fw_off(rr, ch * 8));
        }
        // ---- dWqkv over the stage's 64 rows: output tiles 3 wave .. 3 wave + 2, 5 column tiles (64 + the ones)
#pragma unroll
        for (int ks = 0; ks < FW_ROWS / 32; ks++) {
            bf16x8_t bn[5];
#pragma unroll
            for (int b = 0; b < 5; b++) bn[b] = wg_frag_tr(sN, b, ks, lane);
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const bf16x8_t af = wg_frag_tr(sDQ, 3 * wave + a, ks, lane);
#pragma unroll
                for (int b = 0; b < 5; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bn[b], acc[a][b], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 5; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = (3 * wave + a) * 16 + 4 * g + r, i = b * 16 + lr;
                const float v = acc[a][b][r];
                if (v != 0.0f) { if (i < 64) atomicAdd(&dw[o * 64 + i], v); else if (i == 64) atomicAdd(&dbias[o], v); }
            }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        aw[t] += __shfl_xor(aw[t], 16); aw[t] += __shfl_xor(aw[t], 32);
        ab[t] += __shfl_xor(ab[t], 16); ab[t] += __shfl_xor(ab[t], 32);
    }
    if (g == 0) {
#pragma unroll
        for (int t = 0; t < 4; t++) { atomicAdd(&sG[0][16 * t + lr], aw[t]); atomicAdd(&sG[1][16 * t + lr], ab[t]); }
    }
    __syncthreads();
    if (tid < 64) { atomicAdd(dlnw + tid, sG[0][tid]); atomicAdd(dlnb + tid, sG[1][tid]); }
}

}  // namespace catan
