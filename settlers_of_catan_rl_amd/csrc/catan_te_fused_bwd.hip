// catan_te_fused_bwd.hip - the tile encoder's BACKWARD (reference RL/models/tile_encoder.py:41-91, multi_headed_attention.py under
// RL/ppo/ppo.py:66's loss.backward()) with the forward RECOMPUTED on chip: one kernel per transformer layer, nothing but the layer's
// input per token read from HBM, nothing but the gradient of that input written.
//
// The sub-layer kernels of catan_te_bwd.hip (k_ffn_bwd_w, k_attn_mfma_bwd, k_qkv_bwd_w) walk back through activations that the training
// forward stored: 2.4 KB per token (48 KB per board, 10.6 GB per minibatch step written and read again), and each of them holds so much of
// a 64-row stage in LDS and registers (119 KB, 318 VGPRs) that a CU runs ONE wave per SIMD - every LDS and MFMA latency of the chain
// is exposed.  Here a workgroup of 8 waves (two per SIMD) takes 5 boards = 95 tokens through a whole layer out of LDS, as the
// inference kernel k_tile_encoder_fwd does for the forward:
//   k_te_bwd_layer<1>: xin1 (the input of layer 1: the ONE activation the training forward stores, 128 B per token) and dOut in;
//                      recomputes layer 1, the output projection and the final LayerNorm; walks back; writes d(xin1)
//   k_te_bwd_layer<0>: the tile features and d(xin1) in; recomputes the first layer and layer 0; walks back through both
// Every product runs on v_mfma_f32_16x16x32_bf16 from row-major bf16 tiles in LDS: the forward's products and the dX-type products
// of the backward in te_gemm's transposed form (weights - forward pack or transposed images - as the A operand from registers, fetched
// from L2 one phase ahead); the weight gradients dW = dY^T X contract over the 96 token rows, both operands through ds_read_tr16_b64
// (a 16-lane group reads a [4 rows][16 columns] block and lane i gets column i), accumulated in registers over ALL the groups a
// workgroup takes (persistent workgroups: one per CU) and added to the fp32 gradient block by atomics at the end.  Bias and LayerNorm
// weight / bias gradients are column sums over the tokens: lane = column pair, a few tokens each, two registers per sum.
// Rounding follows the sub-layer chain: every tensor that chain stored in bf16 (dH, dN, dO, dQKV, the residual-stream gradient) is a
// bf16 tile here.  LDS: 8 tiles of 96 rows (149 KB) + the packed fp32 vectors + statistics = 156 KB: one workgroup per CU.
#pragma once

namespace catan {

constexpr int TB_THREADS = 512, TB_W = TB_THREADS / 64;
constexpr int TB_PX = 72, TB_PQ = 200, TB_PH = 144;          // row pitches (bf16 elements) of the 64 / 192 / 128 wide tiles
// the fp32 gradient block of a layer kernel (floats): the layer's parameters, then the kernel's extra ones
constexpr int TG_WQ = 0, TG_BQ = TG_WQ + 192 * 64, TG_WO = TG_BQ + 192, TG_BO = TG_WO + 64 * 64, TG_W1 = TG_BO + 64, TG_B1 = TG_W1 + 128 * 64,
              TG_W2 = TG_B1 + 128, TG_B2 = TG_W2 + 64 * 128, TG_L1W = TG_B2 + 64, TG_L1B = TG_L1W + 64, TG_L2W = TG_L1B + 64, TG_L2B = TG_L2W + 64,
              TG_LAYER = TG_L2B + 64;
// layer 1: out_proj weight [32][64] (rows 25..31 unused), bias [32], final LayerNorm weight / bias [32]
constexpr int TG_WP = TG_LAYER, TG_BP = TG_WP + 32 * 64, TG_LPW = TG_BP + 32, TG_LPB = TG_LPW + 32, TG_TOTAL1 = TG_LPB + 32;
// layer 0: first_layer weight [64][64] (columns 60..63 unused), bias [64], its LayerNorm weight / bias [64]
constexpr int TG_W0 = TG_LAYER, TG_B0 = TG_W0 + 64 * 64, TG_L0W = TG_B0 + 64, TG_L0B = TG_L0W + 64, TG_TOTAL0 = TG_L0B + 64;

struct TeBwdArgs {
    const unsigned short* wts;       // the forward's packed weights (tile_encoder_pack)
    const float* vecs;               // ... and fp32 vectors
    const unsigned short* wqt;       // this layer's transposed images: Wqkv^T [64][192], Wo^T [64][64], W1^T [64][128], W2^T [128][64]
    const unsigned short* wot;
    const unsigned short* w1t;
    const unsigned short* w2t;
    const unsigned short* wpt;       // layer 1: out_proj^T [64][32] (columns 25..31 zero)
    const unsigned short* in_rows;   // layer 1: xin1 [tokens][64]; layer 0: tile features [boards][19][60]
    const unsigned short* dgrad;     // layer 1: dOut [boards][out_pitch]; layer 0: d(xin1) [tokens][64]
    unsigned short* dx_out;          // layer 1: d(xin1) [tokens][64]
    float* G;                        // the gradient block (zeroed by the caller; accumulated by atomics)
    long boards, out_pitch;
};

template <int K, int N> struct TbW { bf16x8_t f[(N / 16 + 3) / 4][K / 32]; };
// this wave's column tiles nt = wave / 2, + 4, ... of a [N][K] row-major bf16 matrix as A-operand fragments
template <int K, int N>
DEVI void tb_fetch(TbW<K, N>& w, const unsigned short* __restrict__ W, int lane, int wave) {
    const int lr = lane & 15, lk = (lane >> 4) * 8, ntb = wave >> 1;
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
// out[m][n] = sum_k A[m][k] W[n][k] (+ bias[n]) over the 96 rows; wave -> column tiles wave / 2 + 4 i, row tiles 3 (wave & 1) .. + 2.
// MODE 0: + bias, store; 1: + bias, ReLU, store; 3: + bias + res[m][n], store; 4: store (no bias); 5: no bias, where out[m][n] (a
// ReLU output) is > 0, else 0, in place (the masked gradient of the hidden layer)
template <int K, int N, int MODE>
DEVI void tb_gemm(const unsigned short* A, int pa, const TbW<K, N>& w, const float* bias, unsigned short* out, int po,
                  const unsigned short* res, int pr, int lane, int wave) {
    constexpr int KS = K / 32, NT = N / 16;
    const int lr = lane & 15, g = lane >> 4, ntb = wave >> 1, m0 = (wave & 1) * 3;
#pragma unroll
    for (int i = 0; i < (NT + 3) / 4; i++) {
        const int nt = ntb + 4 * i;
        if (nt >= NT) break;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE <= 3) bv = *reinterpret_cast<const float4*>(bias + nt * 16 + 4 * g);
#pragma unroll
        for (int mj = 0; mj < 3; mj++) {
            const int row = (m0 + mj) * 16 + lr;
            unsigned short* o = out + row * po + nt * 16 + 4 * g;
            f32x4_t acc = f32x4_t{bv.x, bv.y, bv.z, bv.w};
            if (MODE == 3) {
                const uint2 old = *reinterpret_cast<const uint2*>(res + row * pr + nt * 16 + 4 * g);
                acc[0] += __uint_as_float(old.x << 16); acc[1] += __uint_as_float(old.x & 0xFFFF0000u);
                acc[2] += __uint_as_float(old.y << 16); acc[3] += __uint_as_float(old.y & 0xFFFF0000u);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const uint4 u = *reinterpret_cast<const uint4*>(A + row * pa + ks * 32 + g * 8);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.f[i][ks], *reinterpret_cast<const bf16x8_t*>(&u), acc, 0, 0, 0);
            }
            float v[4] = { acc[0], acc[1], acc[2], acc[3] };
            if (MODE == 1) {
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], 0.f);
            }
            if (MODE == 5) {
                const uint2 h = *reinterpret_cast<const uint2*>(o);
                v[0] = __uint_as_float(h.x << 16) > 0.f ? v[0] : 0.f; v[1] = __uint_as_float(h.x & 0xFFFF0000u) > 0.f ? v[1] : 0.f;
                v[2] = __uint_as_float(h.y << 16) > 0.f ? v[2] : 0.f; v[3] = __uint_as_float(h.y & 0xFFFF0000u) > 0.f ? v[3] : 0.f;
            }
            *reinterpret_cast<uint2*>(o) = make_uint2(pk_bf(v[0], v[1]), pk_bf(v[2], v[3]));
        }
    }
}
// LayerNorm over 64 columns, four lanes per token (te_layer_norm's arithmetic)
template <bool RELU>
DEVI void tb_ln64(const unsigned short* src, int ps, unsigned short* dst, int pd, const float* w, const float* b, int tid) {
    const int t = tid >> 2, e0 = (tid & 3) * 16;
    if (t >= TE_TOK) return;
    float x[16], mean = 0.f;
    te_load8(src + t * ps + e0, x); te_load8(src + t * ps + e0 + 8, x + 8);
#pragma unroll
    for (int i = 0; i < 16; i++) mean += x[i];
    mean += __shfl_xor(mean, 1); mean += __shfl_xor(mean, 2);
    mean *= 1.f / 64.f;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) { const float c = x[i] - mean; var += c * c; }
    var += __shfl_xor(var, 1); var += __shfl_xor(var, 2);
    const float rstd = rsqrtf(var * (1.f / 64.f) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
        const float4 wv = *reinterpret_cast<const float4*>(w + e0 + i), bv = *reinterpret_cast<const float4*>(b + e0 + i);
        const float ww[4] = { wv.x, wv.y, wv.z, wv.w }, bb[4] = { bv.x, bv.y, bv.z, bv.w };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float y = (x[i + k] - mean) * rstd * ww[k] + bb[k];
            if (RELU) y = fmaxf(y, 0.f);
            x[i + k] = y;
        }
    }
    te_store8(dst + t * pd + e0, x); te_store8(dst + t * pd + e0 + 8, x + 8);
}
// Backward of y = LayerNorm64(x) w + b for one token per four lanes: dx = rstd (gw - mean(gw) - xhat mean(gw xhat)), gw = dy w.
// RES:  out = bf(bf(dx) + out)   (the residual-stream gradient, in place; dead rows stay as they are: zero)
// !RES: dy = dyb where y > 0 (the LayerNorm is followed by a ReLU), out = bf(dx); dead rows are written as zeros
// stat[2 t], stat[2 t + 1] = mean, rstd (tb_ln64_colgrad reads them).
template <bool RES>
DEVI void tb_ln64_bwd(const unsigned short* xs, int px, const unsigned short* dyb, int pdy, unsigned short* out, int po,
                      const float* w, const float* b, float* stat, int nt, int tid) {
    const int t = tid >> 2, e0 = (tid & 3) * 16;
    if (t >= TE_ROWS) return;
    if (t >= nt) {
        if (!RES) { const uint4 z = make_uint4(0, 0, 0, 0); *reinterpret_cast<uint4*>(out + t * po + e0) = z; *reinterpret_cast<uint4*>(out + t * po + e0 + 8) = z; }
        return;
    }
    float x[16], gy[16], mean = 0.f;
    te_load8(xs + t * px + e0, x); te_load8(xs + t * px + e0 + 8, x + 8);
    te_load8(dyb + t * pdy + e0, gy); te_load8(dyb + t * pdy + e0 + 8, gy + 8);
#pragma unroll
    for (int i = 0; i < 16; i++) mean += x[i];
    mean += __shfl_xor(mean, 1); mean += __shfl_xor(mean, 2);
    mean *= 1.f / 64.f;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) { x[i] -= mean; var += x[i] * x[i]; }
    var += __shfl_xor(var, 1); var += __shfl_xor(var, 2);
    const float rstd = rsqrtf(var * (1.f / 64.f) + 1e-5f);
    if ((tid & 3) == 0) { stat[2 * t] = mean; stat[2 * t + 1] = rstd; }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        x[i] *= rstd;                                                  // xhat
        const float wi = w[e0 + i];
        if (!RES) { if (!(x[i] * wi + b[e0 + i] > 0.f)) gy[i] = 0.f; }
        gy[i] *= wi;                                                   // gw
        s1 += gy[i]; s2 += gy[i] * x[i];
    }
    s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2);
    s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2);
    const float m1 = s1 * (1.f / 64.f), m2 = s2 * (1.f / 64.f);
    float r[16];
    if (RES) { te_load8(out + t * po + e0, r); te_load8(out + t * po + e0 + 8, r + 8); }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const float d = rstd * (gy[i] - m1 - x[i] * m2);
        x[i] = RES ? hd_bf(d) + r[i] : d;
    }
    te_store8(out + t * po + e0, x); te_store8(out + t * po + e0 + 8, x + 8);
}
// LayerNorm weight / bias gradients as column sums: lane = column pair (tid & 31), tokens tid >> 5, + 16, ...
//   aw[c] += dy[t][c] xhat[t][c], ab[c] += dy[t][c]     (RELU: dy where y > 0)
// (the sums live in LDS - cw[64], cb[64], fp32 - not in registers: 26 more live registers per lane were 26 more spilled ones)
template <bool RELU>
DEVI void tb_ln64_colgrad(const unsigned short* xs, int px, const unsigned short* dyb, int pdy, const float* w, const float* b, const float* stat,
                          int nt, int tid, float* cw, float* cb) {
    const int c = (tid & 31) * 2;
    const float w0 = w[c], w1 = w[c + 1], b0 = b[c], b1 = b[c + 1];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = tid >> 5; t < nt; t += TB_THREADS / 32) {
        const unsigned ux = *reinterpret_cast<const unsigned*>(xs + t * px + c), ud = *reinterpret_cast<const unsigned*>(dyb + t * pdy + c);
        const float2 st = *reinterpret_cast<const float2*>(stat + 2 * t);
        const float x0 = (__uint_as_float(ux << 16) - st.x) * st.y, x1 = (__uint_as_float(ux & 0xFFFF0000u) - st.x) * st.y;
        float d0 = __uint_as_float(ud << 16), d1 = __uint_as_float(ud & 0xFFFF0000u);
        if (RELU) { if (!(x0 * w0 + b0 > 0.f)) d0 = 0.f; if (!(x1 * w1 + b1 > 0.f)) d1 = 0.f; }
        acc[0] += d0 * x0; acc[1] += d1 * x1; acc[2] += d0; acc[3] += d1;
    }
    atomicAdd(cw + c, acc[0]); atomicAdd(cw + c + 1, acc[1]); atomicAdd(cb + c, acc[2]); atomicAdd(cb + c + 1, acc[3]);
}
// column sums of a W-wide tile (bias gradients): lane = column pair
template <int W>
DEVI void tb_colsum(const unsigned short* T, int pitch, int nt, int tid, float* cs) {
    constexpr int PAIRS = W / 2, GROUPS = TB_THREADS / PAIRS;
    const int cp = tid % PAIRS, grp = tid / PAIRS;
    if (grp >= GROUPS) return;
    float acc[2] = {0.f, 0.f};
    for (int t = grp; t < nt; t += GROUPS) {
        const unsigned u = *reinterpret_cast<const unsigned*>(T + t * pitch + 2 * cp);
        acc[0] += __uint_as_float(u << 16); acc[1] += __uint_as_float(u & 0xFFFF0000u);
    }
    atomicAdd(cs + 2 * cp, acc[0]); atomicAdd(cs + 2 * cp + 1, acc[1]);
}
// A / B operand of a product that contracts over the tile's ROWS (k-step of 32 rows from row0): column col0 + (lane & 15), the eight
// rows 4 g + {0..3} and 16 + 4 g + {0..3} (g = lane >> 4) - the permutation of the contraction index both operands share (wg_frag_tr)
DEVI bf16x8_t tb_frag_tr(const unsigned short* T, int pitch, int row0, int col0, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const unsigned short* p = T + (row0 + 4 * g + (i >> 2)) * pitch + col0 + (i & 3) * 4;
    union { bf16x8_t f; wg_s4 h[2]; } r;
    r.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s4*)p);
    r.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s4*)(p + 16 * pitch));
    return r.f;
}
// acc[a][b] += dY[:, ncol0 + 16 a ..]^T X[:, kcol0 + 16 b ..] over the 96 rows
template <int NA, int NB>
DEVI void tb_wgrad(const unsigned short* dY, int py, int ncol0, const unsigned short* X, int px, int kcol0, f32x4_t (&acc)[NA][NB], int lane) {
#pragma unroll
    for (int ks = 0; ks < TE_ROWS / 32; ks++) {
        bf16x8_t bf[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) bf[b] = tb_frag_tr(X, px, 32 * ks, kcol0 + 16 * b, lane);
#pragma unroll
        for (int a = 0; a < NA; a++) {
            const bf16x8_t af = tb_frag_tr(dY, py, 32 * ks, ncol0 + 16 * a, lane);
#pragma unroll
            for (int b = 0; b < NB; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[b], acc[a][b], 0, 0, 0);
        }
    }
}
template <int NA, int NB>
DEVI void tb_wgrad_out(float* __restrict__ g, int ld, int n0, int k0, const f32x4_t (&acc)[NA][NB], int lane) {
#pragma unroll
    for (int a = 0; a < NA; a++)
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float v = acc[a][b][r];
                if (v != 0.f) atomicAdd(g + (long)(n0 + 16 * a + 4 * (lane >> 4) + r) * ld + k0 + 16 * b + (lane & 15), v);
            }
}
// te_attention for eight waves, qkv rows [token][3][head][16] (pitch TB_PQ) -> out rows [token][head * 16 + d] (pitch TB_PX)
DEVI void tb_attention(const unsigned short* qkv, unsigned short* out, int lane, int wave) {
    const int hf = lane >> 5, c31 = lane & 31;
    const bool rowok = c31 < TE_L;
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int i15 = lane & 15;
    const int vrow0 = 4 * hf + (i15 >> 2), vcol = (i15 & 3) * 4;
    const int vrow1 = 16 + vrow0 < TE_L ? 16 + vrow0 : TE_L - 1;
    for (int bh = wave; bh < TE_G * TE_H; bh += TB_W) {
        const int g = bh >> 2, h = bh & 3;
        const unsigned short* base = qkv + g * TE_L * TB_PQ + h * TE_HD;
        const bf16x8_t ka = ld_frag<TE_HD>(base + c31 * TB_PQ + TE_D, hf, rowok);
        const bf16x8_t qb = ld_frag<TE_HD>(base + c31 * TB_PQ, hf, rowok);
        const unsigned short* vb = base + 2 * TE_D + vcol;
        union { bf16x8_t f; wg_s4 q[2]; } va[2];
        va[0].q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s4*)(vb + vrow0 * TB_PQ));
        va[0].q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s4*)(vb + (vrow0 + 8) * TB_PQ));
        va[1].q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s4*)(vb + vrow1 * TB_PQ));
        va[1].q[1] = va[1].q[0];
        const f32x16_t st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qb, zero16, 0, 0, 0);     // S^T[j][i]
        float p[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 12; r++) {
            const int j = (r & 3) + 8 * (r >> 2) + 4 * hf;
            p[r] = j < TE_L ? st[r] : -INFINITY;
            mx = fmaxf(mx, p[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        constexpr float C = 0.25f * 1.44269504088896340736f;
        const float mc = -mx * C;
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 12; r++) { p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(p[r], C, mc)); sum += p[r]; }
        sum += __shfl_xor(sum, 32);
        const float inv = __builtin_amdgcn_rcpf(sum);
        f32x16_t ot = zero16;
#pragma unroll
        for (int s = 0; s < 2; s++)
            ot = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c31 < TE_HD ? va[s].f : zero_bf8(), pack_bf8(p + 8 * s), ot, 0, 0, 0);   // O^T[d][i]
#pragma unroll
        for (int r = 0; r < 8; r++) ot[r] *= inv;
        if (rowok) st_head<TE_HD>(out + (g * TE_L + c31) * TB_PX + h * TE_HD, ot, hf);
    }
}
// rows 16 s + 4 hf + {0..3} and 16 s + 8 + 4 hf + {0..3} of one head's 16 columns of a board's token rows, transposed (lane i & 15 =
// dim): the A operand of the products that contract over keys / queries (k_attn_mfma_bwd's ld_gather_tr).  Rows >= 19 belong to the
// next board (or lie beyond the tile): their coefficients are exact zeros, so they are clamped to the board's last row.
DEVI bf16x8_t tb_gather_tr(const unsigned short* head, int rp, int s, int lane) {
    const int i = lane & 15, hf = lane >> 5;
    int r0 = 16 * s + 4 * hf + (i >> 2), r1 = r0 + 8;
    r0 = r0 < TE_L ? r0 : TE_L - 1; r1 = r1 < TE_L ? r1 : TE_L - 1;
    const unsigned short* p = head + (i & 3) * 4;
    union { bf16x8_t f; wg_s4 h[2]; } r;
    r.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s4*)(p + r0 * rp));
    r.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s4*)(p + r1 * rp));
    return r.f;
}
// k_attn_mfma_bwd on the LDS tiles: dQ | dK | dV replace Q | K | V of the same (board, head) in place; d_o: the gradient of the
// attention output [token][64] (pitch TB_PX).  stat: this wave's 96 floats.
DEVI void tb_attention_bwd(unsigned short* qkv, const unsigned short* d_o, float* stat, int nb, int lane, int wave) {
    const int hf = lane >> 5, c31 = lane & 31;
    const bool rowok = c31 < TE_L;
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int rrow = rowok ? c31 : 0;
    constexpr float scale = 0.25f, C = scale * 1.44269504088896340736f;
    for (int bh = wave; bh < nb * TE_H; bh += TB_W) {
        const int g = bh >> 2, h = bh & 3;
        unsigned short* base = qkv + g * TE_L * TB_PQ + h * TE_HD;
        const unsigned short* dob = d_o + g * TE_L * TB_PX + h * TE_HD;
        const bf16x8_t qr = ld_frag<TE_HD>(base + rrow * TB_PQ, hf, rowok);
        const bf16x8_t kr = ld_frag<TE_HD>(base + rrow * TB_PQ + TE_D, hf, rowok);
        const bf16x8_t vr = ld_frag<TE_HD>(base + rrow * TB_PQ + 2 * TE_D, hf, rowok);
        const bf16x8_t gr = ld_frag<TE_HD>(dob + rrow * TB_PX, hf, rowok);
        f32x16_t dq;
        {   // ---- lane = query i, registers = keys j
            const f32x16_t st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr, qr, zero16, 0, 0, 0);    // S^T[j][i]
            const f32x16_t dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr, gr, zero16, 0, 0, 0);   // dP^T[j][i]
            float p[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 12; r++) {
                const int j = (r & 3) + 8 * (r >> 2) + 4 * hf;
                p[r] = j < TE_L ? st[r] : -INFINITY;
                mx = fmaxf(mx, p[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mc = -mx * C;
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 12; r++) { p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(p[r], C, mc)); sum += p[r]; }
            sum += __shfl_xor(sum, 32);
            const float inv = __builtin_amdgcn_rcpf(sum);
            float delta = 0.f;
#pragma unroll
            for (int r = 0; r < 12; r++) { p[r] *= inv; delta += p[r] * dpt[r]; }
            delta += __shfl_xor(delta, 32);
            if (hf == 0) { stat[c31] = mc; stat[32 + c31] = inv; stat[64 + c31] = delta; }
#pragma unroll
            for (int r = 0; r < 12; r++) p[r] = p[r] * (dpt[r] - delta);                              // dS^T[j][i] / scale
            dq = zero16;
#pragma unroll
            for (int s = 0; s < 2; s++)
                dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c31 < TE_HD ? tb_gather_tr(base + TE_D, TB_PQ, s, lane) : zero_bf8(), pack_bf8(p + 8 * s), dq, 0, 0, 0);   // dQ^T[d][i]
#pragma unroll
            for (int r = 0; r < 8; r++) dq[r] *= scale;
        }
        __builtin_amdgcn_wave_barrier();
        {   // ---- lane = key j, registers = queries i
            const f32x16_t s2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qr, kr, zero16, 0, 0, 0);     // S[i][j]
            const f32x16_t dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gr, vr, zero16, 0, 0, 0);     // dP[i][j]
            float p[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ds[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q4 = 0; q4 < 3; q4++) {                         // registers 4 q4 .. + 3 <-> queries 8 q4 + 4 hf .. + 3
                const float4 m4 = *reinterpret_cast<const float4*>(stat + 8 * q4 + 4 * hf);
                const float4 i4 = *reinterpret_cast<const float4*>(stat + 32 + 8 * q4 + 4 * hf);
                const float4 d4 = *reinterpret_cast<const float4*>(stat + 64 + 8 * q4 + 4 * hf);
                const float mm[4] = { m4.x, m4.y, m4.z, m4.w }, ii[4] = { i4.x, i4.y, i4.z, i4.w }, dd[4] = { d4.x, d4.y, d4.z, d4.w };
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int r = 4 * q4 + e;
                    const bool qok = 8 * q4 + 4 * hf + e < TE_L;     // the rows behind the board's 19 queries are another board's: coefficient 0
                    p[r] = qok ? __builtin_amdgcn_exp2f(__builtin_fmaf(s2[r], C, mm[e])) * ii[e] : 0.f;
                    ds[r] = p[r] * (dp[r] - dd[e]);
                }
            }
            f32x16_t dk = zero16, dv = zero16;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c31 < TE_HD ? tb_gather_tr(base, TB_PQ, s, lane) : zero_bf8(), pack_bf8(ds + 8 * s), dk, 0, 0, 0);   // dK^T[d][j]
                dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c31 < TE_HD ? tb_gather_tr(dob, TB_PX, s, lane) : zero_bf8(), pack_bf8(p + 8 * s), dv, 0, 0, 0);     // dV^T[d][j]
            }
#pragma unroll
            for (int r = 0; r < 8; r++) dk[r] *= scale;
            __builtin_amdgcn_wave_barrier();                         // every read of the head's Q / K / V columns precedes the stores
            if (rowok) {
                st_head<TE_HD>(base + c31 * TB_PQ, dq, hf);
                st_head<TE_HD>(base + c31 * TB_PQ + TE_D, dk, hf);
                st_head<TE_HD>(base + c31 * TB_PQ + 2 * TE_D, dv, hf);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}
// The final LayerNorm(25) + ReLU backward, four lanes per token (8 columns each): P rows (pitch TB_PX, columns 0..31) and the
// gradient of the encoder's output rows (global: token t of board g at dout[g * out_pitch + 25 t ..]) -> dP (bf16) at dst
// (pitch TB_PX; columns 25..31 and the dead rows zero); stat[2 t] = mean, rstd
DEVI void tb_lnp_bwd(const unsigned short* P, unsigned short* dst, const unsigned short* __restrict__ dout, long out_pitch, const float* w, const float* b,
                     float* stat, int nt, int tid) {
    const int t = tid >> 2, e0 = (tid & 3) * 8;
    if (t >= TE_ROWS) return;
    if (t >= nt) { *reinterpret_cast<uint4*>(dst + t * TB_PX + e0) = make_uint4(0, 0, 0, 0); return; }
    float x[8], mean = 0.f;
    te_load8(P + t * TB_PX + e0, x);
#pragma unroll
    for (int i = 0; i < 8; i++) if (e0 + i < TE_OUT) mean += x[i];
    mean += __shfl_xor(mean, 1); mean += __shfl_xor(mean, 2);
    mean *= 1.f / TE_OUT;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) { x[i] -= mean; if (e0 + i < TE_OUT) var += x[i] * x[i]; }
    var += __shfl_xor(var, 1); var += __shfl_xor(var, 2);
    const float rstd = rsqrtf(var * (1.f / TE_OUT) + 1e-5f);
    if ((tid & 3) == 0) { stat[2 * t] = mean; stat[2 * t + 1] = rstd; }
    const int g = t / TE_L;
    const unsigned short* dg = dout + g * out_pitch + (t - g * TE_L) * TE_OUT + e0;
    float gw[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        x[i] *= rstd;
        gw[i] = 0.f;
        if (e0 + i < TE_OUT) {
            const float wi = w[e0 + i];
            const float dy = (x[i] * wi + b[e0 + i] > 0.f) ? te_bf(dg[i]) : 0.f;
            gw[i] = dy * wi;
            s1 += gw[i]; s2 += gw[i] * x[i];
        }
    }
    s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2);
    s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2);
    const float m1 = s1 * (1.f / TE_OUT), m2 = s2 * (1.f / TE_OUT);
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = e0 + i < TE_OUT ? rstd * (gw[i] - m1 - x[i] * m2) : 0.f;
    te_store8(dst + t * TB_PX + e0, x);
}
// ... its weight / bias gradients: lane = column (tid & 31 < 25), tokens tid >> 5, + 16, ...
DEVI void tb_lnp_colgrad(const unsigned short* P, const unsigned short* __restrict__ dout, long out_pitch, const float* w, const float* b, const float* stat,
                         int nt, int tid, float* cw, float* cb) {
    const int c = tid & 31;
    if (c >= TE_OUT) return;
    const float wc = w[c], bc = b[c];
    float acc[2] = {0.f, 0.f};
    for (int t = tid >> 5; t < nt; t += TB_THREADS / 32) {
        const float2 st = *reinterpret_cast<const float2*>(stat + 2 * t);
        const float xh = (te_bf(P[t * TB_PX + c]) - st.x) * st.y;
        const int g = t / TE_L;
        const float dy = (xh * wc + bc > 0.f) ? te_bf(dout[g * out_pitch + (t - g * TE_L) * TE_OUT + c]) : 0.f;
        acc[0] += dy * xh; acc[1] += dy;
    }
    atomicAdd(cw + c, acc[0]); atomicAdd(cb + c, acc[1]);
}

template <int LAYER>
__global__ __launch_bounds__(TB_THREADS) void k_te_bwd_layer(TeBwdArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short XIN[TE_ROWS * TB_PX];     // the layer's input (residual stream)
    __shared__ __attribute__((aligned(16))) unsigned short QKV[TE_ROWS * TB_PQ];     // Q | K | V, then dQ | dK | dV; layer 0's tail: tile features | a0
    __shared__ __attribute__((aligned(16))) unsigned short OB[TE_ROWS * TB_PX];      // attention output O, then dO
    __shared__ __attribute__((aligned(16))) unsigned short XMID[TE_ROWS * TB_PX];    // residual stream behind the attention sub-layer
    __shared__ __attribute__((aligned(16))) unsigned short N2B[TE_ROWS * TB_PX];     // LayerNorm 2 output
    __shared__ __attribute__((aligned(16))) unsigned short HB[TE_ROWS * TB_PH];      // relu(linear1), then dH
    __shared__ __attribute__((aligned(16))) unsigned short DX[TE_ROWS * TB_PX];      // layer output (layer 1), then the residual-stream gradient
    __shared__ __attribute__((aligned(16))) unsigned short TMP[TE_ROWS * TB_PX];     // LayerNorm 1 output / dN / out_proj output and dP / staged input
    __shared__ __attribute__((aligned(16))) float V[TE_VTOTAL];
    __shared__ __attribute__((aligned(16))) float STAT[TE_ROWS * 2];
    __shared__ __attribute__((aligned(16))) float AST[TB_W][96];
    // bias / LayerNorm gradients (column sums): bq [192] b1 [128] bo [64] b2 [64] l1w l1b l2w l2b [64 each] | bx [64] lxw [64] lxb [64]
    constexpr int CS_BQ = 0, CS_B1 = 192, CS_BO = 320, CS_B2 = 384, CS_L1W = 448, CS_L1B = 512, CS_L2W = 576, CS_L2B = 640, CS_BX = 704, CS_LXW = 768, CS_LXB = 832, CS_N = 896;
    __shared__ float CS[CS_N];
    const int tid0 = threadIdx.x, wave = tid0 >> 6;
    {
        const int tid = tid0;
        for (int c = tid; c < TE_VTOTAL / 4; c += TB_THREADS) reinterpret_cast<float4*>(V)[c] = reinterpret_cast<const float4*>(a.vecs)[c];
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (int c = tid; c < TE_ROWS * TB_PX / 8; c += TB_THREADS) {
            reinterpret_cast<uint4*>(XIN)[c] = z; reinterpret_cast<uint4*>(OB)[c] = z; reinterpret_cast<uint4*>(XMID)[c] = z;
            reinterpret_cast<uint4*>(N2B)[c] = z; reinterpret_cast<uint4*>(DX)[c] = z; reinterpret_cast<uint4*>(TMP)[c] = z;
        }
        for (int c = tid; c < TE_ROWS * TB_PQ / 8; c += TB_THREADS) reinterpret_cast<uint4*>(QKV)[c] = z;
        for (int c = tid; c < TE_ROWS * TB_PH / 8; c += TB_THREADS) reinterpret_cast<uint4*>(HB)[c] = z;
    }
    const unsigned short* wl = a.wts + TE_WL + LAYER * TE_WL_SIZE;
    const float* vl = V + TE_VL + LAYER * TE_VL_SIZE;
    // weight-gradient accumulators of this wave (over every group the workgroup takes)
    f32x4_t aq[3][2], ao[1][2], a1[2][2], a2[1][4], ax[1][LAYER == 0 ? 2 : 1];       // ax: dW0 (layer 0) / dWp (layer 1)
    const f32x4_t z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; i++) { aq[i][0] = z4; aq[i][1] = z4; }
    ao[0][0] = z4; ao[0][1] = z4;
#pragma unroll
    for (int i = 0; i < 2; i++) { a1[i][0] = z4; a1[i][1] = z4; }
#pragma unroll
    for (int i = 0; i < 4; i++) a2[0][i] = z4;
    ax[0][0] = z4; ax[0][LAYER == 0 ? 1 : 0] = z4;
    for (int c = tid0; c < CS_N; c += TB_THREADS) CS[c] = 0.f;
    const int wq = wave >> 1, wh = wave & 1;
    const long ngroups = (a.boards + TE_G - 1) / TE_G;
    __syncthreads();
    for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        // the lane index is made opaque once per group: every per-lane LDS address of the ~25 phases below is a function of it, and hoisted
        // out of this loop as loop invariants they were ~200 registers that lived across the whole kernel - and were spilled
        int lane_ = tid0 & 63;
        asm volatile("" : "+v"(lane_));
        const int lane = lane_, tid = wave * 64 + lane;
        const long g0 = grp * TE_G;
        const int nb = (int)(a.boards - g0 < TE_G ? a.boards - g0 : TE_G);
        const int nt = nb * TE_L;
        const long t0 = g0 * TE_L;
        // ================================================================= the layer's input
        if constexpr (LAYER == 1) {
            TbW<64, 192> wq_; tb_fetch<64, 192>(wq_, wl, lane, wave);
            for (int c = tid; c < TE_ROWS * 8; c += TB_THREADS) {
                const int row = c >> 3, ch = c & 7;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (row < nt) v = *reinterpret_cast<const uint4*>(a.in_rows + (t0 + row) * 64 + ch * 8);
                *reinterpret_cast<uint4*>(XIN + row * TB_PX + ch * 8) = v;
            }
            __syncthreads();
            tb_ln64<false>(XIN, TB_PX, TMP, TB_PX, vl, vl + 64, tid);
            __syncthreads();
            tb_gemm<64, 192, 0>(TMP, TB_PX, wq_, vl + 128, QKV, TB_PQ, nullptr, 0, lane, wave);
        } else {
            TbW<64, 64> w0; tb_fetch<64, 64>(w0, a.wts + TE_W0, lane, wave);
            for (int c = tid; c < TE_ROWS * 16; c += TB_THREADS) {
                const int row = c >> 4, ch = c & 15;
                uint2 v = make_uint2(0u, 0u);
                if (ch < 15 && row < nt) v = *reinterpret_cast<const uint2*>(a.in_rows + (t0 + row) * TE_IN + ch * 4);
                *reinterpret_cast<uint2*>(TMP + row * TB_PX + ch * 4) = v;
            }
            __syncthreads();
            tb_gemm<64, 64, 0>(TMP, TB_PX, w0, V + TE_V0, XIN, TB_PX, nullptr, 0, lane, wave);
            TbW<64, 192> wq_; tb_fetch<64, 192>(wq_, wl, lane, wave);
            __syncthreads();
            tb_ln64<true>(XIN, TB_PX, XIN, TB_PX, V + TE_V0 + 64, V + TE_V0 + 128, tid);
            __syncthreads();
            tb_ln64<false>(XIN, TB_PX, TMP, TB_PX, vl, vl + 64, tid);
            __syncthreads();
            tb_gemm<64, 192, 0>(TMP, TB_PX, wq_, vl + 128, QKV, TB_PQ, nullptr, 0, lane, wave);
        }
        // ================================================================= forward of the layer (k_tile_encoder_fwd's phases)
        {
            TbW<64, 64> wo; tb_fetch<64, 64>(wo, wl + 192 * 64, lane, wave);
            __syncthreads();
            tb_attention(QKV, OB, lane, wave);
            __syncthreads();
            tb_gemm<64, 64, 3>(OB, TB_PX, wo, vl + 320, XMID, TB_PX, XIN, TB_PX, lane, wave);
        }
        {
            TbW<64, 128> w1; tb_fetch<64, 128>(w1, wl + 192 * 64 + 64 * 64, lane, wave);
            __syncthreads();
            tb_ln64<false>(XMID, TB_PX, N2B, TB_PX, vl + 384, vl + 448, tid);
            __syncthreads();
            tb_gemm<64, 128, 1>(N2B, TB_PX, w1, vl + 512, HB, TB_PH, nullptr, 0, lane, wave);
        }
        if constexpr (LAYER == 1) {
            // ---- the layer's output, the output projection, the final LayerNorm + ReLU backward
            TbW<128, 64> w2; tb_fetch<128, 64>(w2, wl + 192 * 64 + 64 * 64 + 128 * 64, lane, wave);
            TbW<64, 32> wp; tb_fetch<64, 32>(wp, a.wts + TE_WP, lane, wave);
            __syncthreads();
            tb_gemm<128, 64, 3>(HB, TB_PH, w2, vl + 640, DX, TB_PX, XMID, TB_PX, lane, wave);                 // xfin
            __syncthreads();
            tb_gemm<64, 32, 0>(DX, TB_PX, wp, V + TE_VP, TMP, TB_PX, nullptr, 0, lane, wave);                 // P: columns 0..31 of TMP
            __syncthreads();
            tb_lnp_bwd(TMP, TMP + 32, a.dgrad + g0 * a.out_pitch, a.out_pitch, V + TE_VP + 32, V + TE_VP + 64, STAT, nt, tid);   // dP: columns 32..63
            TbW<32, 64> wpt; tb_fetch<32, 64>(wpt, a.wpt, lane, wave);
            __syncthreads();
            tb_lnp_colgrad(TMP, a.dgrad + g0 * a.out_pitch, a.out_pitch, V + TE_VP + 32, V + TE_VP + 64, STAT, nt, tid, CS + CS_LXW, CS + CS_LXB);
            tb_colsum<32>(TMP + 32, TB_PX, nt, tid, CS + CS_BX);
            tb_wgrad<1, 1>(TMP + 32, TB_PX, 16 * (wave >> 2), DX, TB_PX, 16 * (wave & 3), ax, lane);             // dWp += dP^T xfin
            __syncthreads();
            tb_gemm<32, 64, 4>(TMP + 32, TB_PX, wpt, nullptr, DX, TB_PX, nullptr, 0, lane, wave);             // d(xfin) = dP Wp
        } else {
            for (int c = tid; c < TE_ROWS * 8; c += TB_THREADS) {                                             // d(xin1) from the layer-1 kernel
                const int row = c >> 3, ch = c & 7;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (row < nt) v = *reinterpret_cast<const uint4*>(a.dgrad + (t0 + row) * 64 + ch * 8);
                *reinterpret_cast<uint4*>(DX + row * TB_PX + ch * 8) = v;
            }
        }
        // ================================================================= backward: FFN sub-layer
        {
            TbW<64, 128> w2t; tb_fetch<64, 128>(w2t, a.w2t, lane, wave);
            __syncthreads();
            tb_wgrad<1, 4>(DX, TB_PX, 16 * wq, HB, TB_PH, 64 * wh, a2, lane);                                 // dW2 += dX^T H
            tb_colsum<64>(DX, TB_PX, nt, tid, CS + CS_B2);
            __syncthreads();
            tb_gemm<64, 128, 5>(DX, TB_PX, w2t, nullptr, HB, TB_PH, nullptr, 0, lane, wave);                  // dH = (dX W2) where H > 0, in place
        }
        {
            TbW<128, 64> w1t; tb_fetch<128, 64>(w1t, a.w1t, lane, wave);
            __syncthreads();
            tb_wgrad<2, 2>(HB, TB_PH, 32 * wq, N2B, TB_PX, 32 * wh, a1, lane);                                // dW1 += dH^T N2
            tb_colsum<128>(HB, TB_PH, nt, tid, CS + CS_B1);
            tb_gemm<128, 64, 4>(HB, TB_PH, w1t, nullptr, TMP, TB_PX, nullptr, 0, lane, wave);                 // dN2 = dH W1
            __syncthreads();
            tb_ln64_bwd<true>(XMID, TB_PX, TMP, TB_PX, DX, TB_PX, vl + 384, vl + 448, STAT, nt, tid);         // d(xmid) = dX + LayerNorm'(dN2)
        }
        // ================================================================= backward: attention sub-layer
        {
            TbW<64, 64> wot; tb_fetch<64, 64>(wot, a.wot, lane, wave);
            __syncthreads();
            tb_ln64_colgrad<false>(XMID, TB_PX, TMP, TB_PX, vl + 384, vl + 448, STAT, nt, tid, CS + CS_L2W, CS + CS_L2B);
            tb_wgrad<1, 2>(DX, TB_PX, 16 * wq, OB, TB_PX, 32 * wh, ao, lane);                                 // dWo += d(xmid)^T O
            tb_colsum<64>(DX, TB_PX, nt, tid, CS + CS_BO);
            __syncthreads();
            tb_gemm<64, 64, 4>(DX, TB_PX, wot, nullptr, OB, TB_PX, nullptr, 0, lane, wave);                   // dO = d(xmid) Wo
            __syncthreads();
            tb_attention_bwd(QKV, OB, AST[wave], nb, lane, wave);                                             // dQKV in place
            for (int c = tid; c < (TE_ROWS - nt) * (192 / 8); c += TB_THREADS) {                              // the rows behind the last token: zero
                const int row = nt + c / 24, ch = c % 24;
                *reinterpret_cast<uint4*>(QKV + row * TB_PQ + ch * 8) = make_uint4(0, 0, 0, 0);
            }
            tb_ln64<false>(XIN, TB_PX, TMP, TB_PX, vl, vl + 64, tid);                                         // N1 again (the QKV product's input)
        }
        {
            TbW<192, 64> wqt; tb_fetch<192, 64>(wqt, a.wqt, lane, wave);
            __syncthreads();
            tb_wgrad<3, 2>(QKV, TB_PQ, 48 * wq, TMP, TB_PX, 32 * wh, aq, lane);                               // dWqkv += dQKV^T N1
            tb_colsum<192>(QKV, TB_PQ, nt, tid, CS + CS_BQ);
            __syncthreads();
            tb_gemm<192, 64, 4>(QKV, TB_PQ, wqt, nullptr, TMP, TB_PX, nullptr, 0, lane, wave);                // dN1 = dQKV Wqkv
            __syncthreads();
            tb_ln64_bwd<true>(XIN, TB_PX, TMP, TB_PX, DX, TB_PX, vl, vl + 64, STAT, nt, tid);                 // d(xin) = d(xmid) + LayerNorm'(dN1)
            __syncthreads();
            tb_ln64_colgrad<false>(XIN, TB_PX, TMP, TB_PX, vl, vl + 64, STAT, nt, tid, CS + CS_L1W, CS + CS_L1B);
        }
        if constexpr (LAYER == 1) {
            for (int c = tid; c < nt * 8; c += TB_THREADS) {
                const int row = c >> 3, ch = c & 7;
                *reinterpret_cast<uint4*>(a.dx_out + (t0 + row) * 64 + ch * 8) = *reinterpret_cast<const uint4*>(DX + row * TB_PX + ch * 8);
            }
        } else {
            // ---- the first layer: x0 = relu(LayerNorm(a0)), a0 = tiles W0^T + b0 (both recomputed: their tiles were reused)
            TbW<64, 64> w0; tb_fetch<64, 64>(w0, a.wts + TE_W0, lane, wave);
            for (int c = tid; c < TE_ROWS * 16; c += TB_THREADS) {
                const int row = c >> 4, ch = c & 15;
                uint2 v = make_uint2(0u, 0u);
                if (ch < 15 && row < nt) v = *reinterpret_cast<const uint2*>(a.in_rows + (t0 + row) * TE_IN + ch * 4);
                *reinterpret_cast<uint2*>(QKV + row * TB_PQ + ch * 4) = v;                                    // tile features: columns 0..63 of QKV's rows
            }
            __syncthreads();
            tb_gemm<64, 64, 0>(QKV, TB_PQ, w0, V + TE_V0, QKV + 72, TB_PQ, nullptr, 0, lane, wave);           // a0: columns 72..135
            __syncthreads();
            tb_ln64_bwd<false>(QKV + 72, TB_PQ, DX, TB_PX, TMP, TB_PX, V + TE_V0 + 64, V + TE_V0 + 128, STAT, nt, tid);   // dA0
            __syncthreads();
            tb_ln64_colgrad<true>(QKV + 72, TB_PQ, DX, TB_PX, V + TE_V0 + 64, V + TE_V0 + 128, STAT, nt, tid, CS + CS_LXW, CS + CS_LXB);
            tb_wgrad<1, 2>(TMP, TB_PX, 16 * wq, QKV, TB_PQ, 32 * wh, ax, lane);                               // dW0 += dA0^T tiles
            tb_colsum<64>(TMP, TB_PX, nt, tid, CS + CS_BX);
        }
        __syncthreads();                                                                                     // (the next group's staging overwrites the tiles)
    }
    // ================================================================= the accumulators leave (fp32 atomics; the block was zeroed by the caller)
    const int tid = tid0, lane = tid0 & 63;
    float* G = a.G;
    tb_wgrad_out<3, 2>(G + TG_WQ, 64, 48 * wq, 32 * wh, aq, lane);
    tb_wgrad_out<1, 2>(G + TG_WO, 64, 16 * wq, 32 * wh, ao, lane);
    tb_wgrad_out<2, 2>(G + TG_W1, 64, 32 * wq, 32 * wh, a1, lane);
    tb_wgrad_out<1, 4>(G + TG_W2, 128, 16 * wq, 64 * wh, a2, lane);
    __syncthreads();
    for (int c = tid; c < CS_N; c += TB_THREADS) {
        const float v = CS[c];
        int o = -1;
        if (c < CS_B1) o = TG_BQ + c;
        else if (c < CS_BO) o = TG_B1 + c - CS_B1;
        else if (c < CS_B2) o = TG_BO + c - CS_BO;
        else if (c < CS_L1W) o = TG_B2 + c - CS_B2;
        else if (c < CS_BX) o = TG_L1W + c - CS_L1W;                   // l1w, l1b, l2w, l2b: contiguous in both
        else if (LAYER == 0) o = TG_B0 + c - CS_BX;                    // b0, l0w, l0b: contiguous in both
        else if (c < CS_LXW) { if (c - CS_BX < 32) o = TG_BP + c - CS_BX; }
        else if (c < CS_LXB) { if (c - CS_LXW < 32) o = TG_LPW + c - CS_LXW; }
        else if (c - CS_LXB < 32) o = TG_LPB + c - CS_LXB;
        if (o >= 0 && v != 0.f) atomicAdd(G + o, v);
    }
    if constexpr (LAYER == 1) tb_wgrad_out<1, 1>(G + TG_WP, 64, 16 * (wave >> 2), 16 * (wave & 3), ax, lane);
    else tb_wgrad_out<1, 2>(G + TG_W0, 64, 16 * wq, 32 * wh, ax, lane);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The two ENDS of the encoder's backward as one kernel each, beside the sub-layer kernels of catan_te_bwd.hip (the default chain):
//   k_te_bwd_ends<1>  behind the last transformer layer: out = relu(LayerNorm25(xfin Wp^T + bp)).  xfin [tokens][64] and dOut in; P is
//                     recomputed (one 64 x 32 product per token tile), LayerNorm + ReLU backward, dWp / dbp / LayerNorm gradients, d(xfin) out.
//                     As separate kernels (LayerNorm-25 backward over 50-byte rows, a 25 -> 64 row product, a weight gradient) the chain moved
//                     506 B per token in three passes and needed P stored by the forward; here 306 B in one.
//   k_te_bwd_ends<0>  in front of the first layer: x0 = relu(LayerNorm(tiles W0^T + b0)).  The tile features and d(x0) in; a0 is recomputed,
//                     LayerNorm + ReLU backward, dW0 / db0 / LayerNorm gradients.  (Separately: 640 B per token in two passes and a0 + the padded
//                     tile features stored by the forward; here 248 B in one.)
// Same phase functions as k_te_bwd_layer (95-token groups in LDS, 512 threads); the tiles are few (28 / 55 KB), so several workgroups share a CU.
// G: <1>: Wp [32][64] | bp [32] | LayerNorm weight, bias [32 each];  <0>: W0 [64][64] | b0 [64] | LayerNorm weight, bias [64 each].
constexpr int TGE_W = 0, TGE1_B = 32 * 64, TGE1_LW = TGE1_B + 32, TGE1_LB = TGE1_LW + 32, TGE1_TOTAL = TGE1_LB + 32;
constexpr int TGE0_B = 64 * 64, TGE0_LW = TGE0_B + 64, TGE0_LB = TGE0_LW + 64, TGE0_TOTAL = TGE0_LB + 64;
template <int PART>
__global__ __launch_bounds__(TB_THREADS) void k_te_bwd_ends(TeBwdArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short XA[TE_ROWS * TB_PX];                  // <1>: xfin, then d(xfin);  <0>: the tile features
    __shared__ __attribute__((aligned(16))) unsigned short TMP[TE_ROWS * TB_PX];                 // <1>: P | dP;  <0>: dA0
    __shared__ __attribute__((aligned(16))) unsigned short A0[PART == 0 ? TE_ROWS * TB_PX : 8];  // <0>: a0
    __shared__ __attribute__((aligned(16))) unsigned short DX[PART == 0 ? TE_ROWS * TB_PX : 8];  // <0>: d(x0)
    __shared__ __attribute__((aligned(16))) float V[PART == 1 ? 96 : 192];                       // the part's bias / LayerNorm vectors
    __shared__ __attribute__((aligned(16))) float STAT[TE_ROWS * 2];
    __shared__ float CS[192];                                                                    // bias | LayerNorm weight | LayerNorm bias column sums (64 each)
    const int tid0 = threadIdx.x, wave = tid0 >> 6;
    {
        const float* gv = a.vecs + (PART == 1 ? TE_VP : TE_V0);
        for (int c = tid0; c < (PART == 1 ? 96 : 192); c += TB_THREADS) V[c] = gv[c];
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (int c = tid0; c < TE_ROWS * TB_PX / 8; c += TB_THREADS) {
            reinterpret_cast<uint4*>(XA)[c] = z; reinterpret_cast<uint4*>(TMP)[c] = z;
            if (PART == 0) { reinterpret_cast<uint4*>(A0)[c] = z; reinterpret_cast<uint4*>(DX)[c] = z; }
        }
        for (int c = tid0; c < 192; c += TB_THREADS) CS[c] = 0.f;
    }
    f32x4_t ax[1][PART == 0 ? 2 : 1];
    const f32x4_t z4 = {0.f, 0.f, 0.f, 0.f};
    ax[0][0] = z4; ax[0][PART == 0 ? 1 : 0] = z4;
    const int wq = wave >> 1, wh = wave & 1;
    const long ngroups = (a.boards + TE_G - 1) / TE_G;
    __syncthreads();
    for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        int lane_ = tid0 & 63;
        asm volatile("" : "+v"(lane_));                                  // (see k_te_bwd_layer: keeps the per-lane addresses out of the loop-invariant set)
        const int lane = lane_, tid = wave * 64 + lane;
        const long g0 = grp * TE_G;
        const int nb = (int)(a.boards - g0 < TE_G ? a.boards - g0 : TE_G);
        const int nt = nb * TE_L;
        const long t0 = g0 * TE_L;
        if constexpr (PART == 1) {
            TbW<64, 32> wp; tb_fetch<64, 32>(wp, a.wts + TE_WP, lane, wave);
            TbW<32, 64> wpt; tb_fetch<32, 64>(wpt, a.wpt, lane, wave);
            for (int c = tid; c < TE_ROWS * 8; c += TB_THREADS) {
                const int row = c >> 3, ch = c & 7;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (row < nt) v = *reinterpret_cast<const uint4*>(a.in_rows + (t0 + row) * 64 + ch * 8);
                *reinterpret_cast<uint4*>(XA + row * TB_PX + ch * 8) = v;
            }
            __syncthreads();
            tb_gemm<64, 32, 0>(XA, TB_PX, wp, V, TMP, TB_PX, nullptr, 0, lane, wave);                          // P: columns 0..31 of TMP
            __syncthreads();
            tb_lnp_bwd(TMP, TMP + 32, a.dgrad + g0 * a.out_pitch, a.out_pitch, V + 32, V + 64, STAT, nt, tid);  // dP: columns 32..63
            __syncthreads();
            tb_lnp_colgrad(TMP, a.dgrad + g0 * a.out_pitch, a.out_pitch, V + 32, V + 64, STAT, nt, tid, CS + 64, CS + 128);
            tb_colsum<32>(TMP + 32, TB_PX, nt, tid, CS);
            tb_wgrad<1, 1>(TMP + 32, TB_PX, 16 * (wave >> 2), XA, TB_PX, 16 * (wave & 3), ax, lane);             // dWp += dP^T xfin
            __syncthreads();
            tb_gemm<32, 64, 4>(TMP + 32, TB_PX, wpt, nullptr, XA, TB_PX, nullptr, 0, lane, wave);               // d(xfin) = dP Wp
            __syncthreads();
            for (int c = tid; c < nt * 8; c += TB_THREADS) {
                const int row = c >> 3, ch = c & 7;
                *reinterpret_cast<uint4*>(a.dx_out + (t0 + row) * 64 + ch * 8) = *reinterpret_cast<const uint4*>(XA + row * TB_PX + ch * 8);
            }
        } else {
            TbW<64, 64> w0; tb_fetch<64, 64>(w0, a.wts + TE_W0, lane, wave);
            for (int c = tid; c < TE_ROWS * 16; c += TB_THREADS) {
                const int row = c >> 4, ch = c & 15;
                uint2 v = make_uint2(0u, 0u);
                if (ch < 15 && row < nt) v = *reinterpret_cast<const uint2*>(a.in_rows + (t0 + row) * TE_IN + ch * 4);
                *reinterpret_cast<uint2*>(XA + row * TB_PX + ch * 4) = v;
            }
            for (int c = tid; c < TE_ROWS * 8; c += TB_THREADS) {
                const int row = c >> 3, ch = c & 7;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (row < nt) v = *reinterpret_cast<const uint4*>(a.dgrad + (t0 + row) * 64 + ch * 8);
                *reinterpret_cast<uint4*>(DX + row * TB_PX + ch * 8) = v;
            }
            __syncthreads();
            tb_gemm<64, 64, 0>(XA, TB_PX, w0, V, A0, TB_PX, nullptr, 0, lane, wave);                           // a0
            __syncthreads();
            tb_ln64_bwd<false>(A0, TB_PX, DX, TB_PX, TMP, TB_PX, V + 64, V + 128, STAT, nt, tid);               // dA0
            __syncthreads();
            tb_ln64_colgrad<true>(A0, TB_PX, DX, TB_PX, V + 64, V + 128, STAT, nt, tid, CS + 64, CS + 128);
            tb_wgrad<1, 2>(TMP, TB_PX, 16 * wq, XA, TB_PX, 32 * wh, ax, lane);                                  // dW0 += dA0^T tiles
            tb_colsum<64>(TMP, TB_PX, nt, tid, CS);
        }
        __syncthreads();
    }
    const int lane = tid0 & 63;
    float* G = a.G;
    if constexpr (PART == 1) {
        tb_wgrad_out<1, 1>(G + TGE_W, 64, 16 * (wave >> 2), 16 * (wave & 3), ax, lane);
        if (tid0 < 96 && (tid0 & 31) < 32) { const float v = CS[(tid0 >> 5) * 64 + (tid0 & 31)]; if (v != 0.f) atomicAdd(G + TGE1_B + tid0, v); }
    } else {
        tb_wgrad_out<1, 2>(G + TGE_W, 64, 16 * wq, 32 * wh, ax, lane);
        if (tid0 < 192) { const float v = CS[tid0]; if (v != 0.f) atomicAdd(G + TGE0_B + tid0, v); }
    }
}

}  // namespace catan
