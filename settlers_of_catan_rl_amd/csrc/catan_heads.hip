// catan_heads.hip - one action head of the policy net as ONE inference kernel (acting in rollouts, evaluation, forward search).
//
// A head is  mlp_1 -> LayerNorm -> ReLU -> mlp_2 -> distribution.linear -> masked categorical  (reference
// RL/models/action_heads_module.py:202-228, RL/distributions.py:10-40).  Its mlp_1 over the trunk comes out of the one GEMM all
// twelve heads share (policy._ActionHeads: `pre_all`); what is left per head evaluation was ~8 launches of 5-15 us each - the
// conditioning columns' product and add, LayerNorm + ReLU, two small GEMMs, a cast, the categorical - and a policy pass makes
// twenty such evaluations one after the other (the heads are autoregressive): 4.3 of the 6 ms of a 65 536-row pass.
// Here, per head evaluation: a 256-thread workgroup copies the head's weights to LDS (W2 128x128, W3 Kx128, the conditioning
// columns of W1: 64 KB, from L2) and takes 256 rows through the whole chain, 16 rows per wave pass:
//   x = pre (+ cond . W1e^T)            16-byte loads in the MFMA operand layout: lane = (row l % 16, eight k at 8 (l / 16) + 32 s)
//   y = relu(LayerNorm(x))              in registers; a row's 128 values sit in 4 lanes x 32: two xor-shuffles per statistic
//   h^T = W2 . y^T + b2                 v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A operand: the result leaves a lane with
//                                       h[row l % 16][16 j + 4 (l / 16) + i] - which IS the A-operand layout of the next product
//                                       once its contraction index is permuted (both operands of a product may permute k alike)
//   logits = h . W3^T + b3              B fragments = two 8-byte LDS reads in the same permuted order; K <= 80 (5 column tiles)
//   masked categorical                  logits through a small LDS tile; 4 lanes per row: max / sum / inverse-CDF pick with
//                                       quad shuffles; the first index whose cumulative probability exceeds u, as k_categorical_fwd
// Rounding follows the unfused bf16-autocast path: bf16 after the conditioning add, after LayerNorm + ReLU, after each Linear
// (bias added in fp32 first); softmax statistics in fp32.
#pragma once

namespace catan {

constexpr int HD_PITCH = 136;            // LDS row pitch of the weight matrices (bf16 elements): 272 B, 16-byte aligned, conflict-free fragments
constexpr int HD_LG = 84;                // floats per logits row in LDS (row groups 16 banks apart)
// Row tiles per wave x waves per workgroup.  An evaluation is ~20-25 us of dependent latency whatever the row count (weights to LDS, the
// state / mask round trips, the tile's chain); one workgroup per CU fits in LDS.  WIDE (2 x 8: 256 rows per workgroup, two waves per SIMD)
// covers 65 536 rows with one workgroup per CU; NARROW (1 x 12: 192 rows, three waves per SIMD, one tile each) shortens the chain per wave -
// 440 -> 345 us per eighteen evaluations at 4 096 rows, 490 -> 385 us at 16 384 - but would need 342 workgroups at 65 536 rows (800 us):
// the launcher picks NARROW up to HD_NARROW_MAX_ROWS rows (the forward search's thinning batches, the rollout's tail).
constexpr int HD_RT_WIDE = 2, HD_WAVES_WIDE = 8, HD_RT_NARROW = 1, HD_WAVES_NARROW = 12;
constexpr long HD_NARROW_MAX_ROWS = 49152;                 // 256 workgroups of 192 rows
constexpr int HD_NCP = 32;               // conditioning columns, padded
constexpr int HD_KP = 80;                // output columns, padded (73 road edges)
constexpr int HD_WELEMS = 128 * 128 + HD_KP * 128 + HD_NCP * 128;      // packed bf16: W2 [128][128], W3 [80][128], W1e^T [32][128]
constexpr int HD_VELEMS = 128 * 3 + HD_KP;                             // packed fp32: ln_w, ln_b, b2 [128] each, b3 [80]

DEVI float hd_bf(float v) { return te_bf(te_to_bf(v)); }

struct HeadArgs {
    const unsigned short* pre; long pre_ld;      // bf16 [B][.. 128 ..]: this head's columns of the shared trunk product (bias included)
    const float* cond; long cond_ld; int ncond;  // float [B][ncond] conditioning columns that follow the trunk in mlp_1's input, or null
    const unsigned short* wts; const float* vec; // the head's packs (HD_WELEMS / HD_VELEMS)
    float eps; int K;
    const float* mask; long mask_ld;             // float [B][K] (a column window of the mask matrix is fine)
    const float* u;                              // uniform per row (inverse-CDF sample) or null (arg-max)
    long long* action; float* logp; long B;
    // ---- chained mode (state != null): the autoregressive glue of the twelve heads (which mask row, which conditioning
    // columns, whether the head counts towards the joint log-prob, the trade heads' running hand) happens in the kernel, from a
    // small per-row state that the twenty evaluations of a policy pass hand on to each other; cond / mask / action / logp above
    // are unused then.  Restates RL/models/build_agent_model.py:113-147 + action_heads_module.py:66-179,258-329 per row.
    float* state;                                // [B][HD_STATE]
    int head_id, step;                           // head 0..11; step 0..3 of the recurrent trade heads 7 / 8
    const float* maskmat;                        // float [B][325]: the env's masks
    const float* cur_res;                        // float [B][6]: current_resources (index 0 unused)
    const float* trade;                          // float [B][12]: proposed_trade (head 5)
    const float* custom;                         // head 5: custom_mlp W [32][12], b [32], custom_norm w [32], b [32] (bf16-rounded W, b)
    const long long* forced;                     // head 0: rows with a value >= 0 take that type (condition_on_action_type) or null
    long long* actions;                          // int64 [B][18]
    float* logp_out;                             // [B]: the joint log-prob, written by the last evaluation (head 8, step 3)
};
constexpr int HD_STATE = 32;                     // floats per row
// state slots
constexpr int HS_TYP = 0, HS_CARD = 1, HS_RA = 2, HS_CNT9 = 3, HS_TOTAL = 4, HS_LPSUM = 5, HS_PREV = 6, HS_FILT7 = 7, HS_OUT = 8, HS_RES = 14, HS_GIVE = 20;

// what a row's evaluation of head `h` needs besides its mask: its conditioning columns (-> cd) and the factor its log-prob enters
// the joint log-prob with (log_prob_masks, build_agent_model.py:132-147); the trade heads' running hand starts here.  One lane per row.
DEVI float hd_glue(const HeadArgs& a, long row, float* st, float* cd) {
    const int h = a.head_id;
    const int typ = (int)st[HS_TYP], card = (int)st[HS_CARD];
    auto is = [&](int t) { return typ == t ? 1.0f : 0.0f; };
    switch (h) {
    case 0: return 1.0f;
    case 1: cd[0] = is(T_SETTLE); cd[1] = is(T_CITY); return is(T_SETTLE) + is(T_CITY);
    case 2: return is(T_ROAD);
    case 3: return is(T_ROBBER);
    case 4: return is(T_PLAYDEV);
    case 5: {                                                              // accept / reject: conditioned on the offer (custom_mlp + LayerNorm + ReLU)
        const float* tr = a.trade + row * 12;
        float t[32], mean = 0.0f, var = 0.0f;
        for (int o = 0; o < 32; o++) {
            float acc = 0.0f;
            for (int k = 0; k < 12; k++) acc += a.custom[o * 12 + k] * hd_bf(tr[k]);
            t[o] = hd_bf(acc + a.custom[384 + o]);
            mean += t[o];
        }
        mean *= 1.0f / 32.0f;
        for (int o = 0; o < 32; o++) var += (t[o] - mean) * (t[o] - mean);
        const float rstd = rsqrtf(var * (1.0f / 32.0f) + a.eps);
        for (int o = 0; o < 32; o++) cd[o] = hd_bf(fmaxf((t[o] - mean) * rstd * a.custom[416 + o] + a.custom[448 + o], 0.0f));
        return is(T_RESPOND);
    }
    case 6: cd[0] = is(T_PROPOSE); cd[1] = is(T_STEAL); return is(T_PROPOSE) + is(T_STEAL);
    case 9: case 10: {                                                     // resource A / B of an exchange, Year of Plenty or Monopoly
        const bool playdev = typ == T_PLAYDEV;
        cd[0] = is(T_PLAYDEV); cd[1] = is(T_EXCHANGE);
        cd[2] = (playdev && card == C_YOP) ? 1.0f : 0.0f; cd[3] = (playdev && card == C_MONO) ? 1.0f : 0.0f;
        const float base = is(T_PLAYDEV) + is(T_EXCHANGE);
        if (h == 9) return base * (playdev ? ((card == C_YOP || card == C_MONO) ? 1.0f : 0.0f) : 1.0f);
        const int ra = (int)st[HS_RA];
        for (int k = 0; k < 5; k++) cd[4 + k] = (k == ra && st[HS_CNT9] != 0.0f) ? 1.0f : 0.0f;
        return base * (playdev ? (card == C_YOP ? 1.0f : 0.0f) : 1.0f);
    }
    case 11: return is(T_DISCARD);
    default: {                                                             // 7: give list (from the hand), 8: receive list; four steps each
        if (a.step == 0) {
            for (int k = 0; k < 6; k++) { st[HS_OUT + k] = 0.0f; st[HS_RES + k] = a.cur_res[row * 6 + k]; }
            st[HS_LPSUM] = 0.0f; st[HS_PREV] = 1.0f;
        }
        if (h == 7) { for (int k = 0; k < 6; k++) cd[k] = st[HS_OUT + k]; }
        else { for (int k = 0; k < 6; k++) { cd[k] = st[HS_GIVE + k] * (1.0f - st[HS_FILT7]); cd[6 + k] = st[HS_OUT + k]; } }
        return 1.0f;
    }
    }
}
// the row's mask entries of columns c0 .. c0 + 19 (the categorical's split: four lanes per row), after hd_glue: which row of the
// env's mask matrix depends on the type / card the earlier heads chose; the trade lists' masks are the running hand
DEVI void hd_mask20(const HeadArgs& a, long row, const float* st, int c0, float* mk) {
    const float* mm = a.maskmat + row * 325;
    const int h = a.head_id;
    const int typ = (int)st[HS_TYP], card = (int)st[HS_CARD];
    const float* p1 = nullptr; const float* p2 = nullptr;
    switch (h) {
    case 0: p1 = mm + M0; break;
    case 1: p1 = mm + M1 + 54 * (typ == T_SETTLE ? 0 : (typ == T_CITY ? 1 : 2)); break;   // settlement row, city row or the dummy row
    case 2: p1 = mm + M2; break;
    case 3: p1 = mm + M3; break;
    case 4: p1 = mm + M4; break;
    case 5: p1 = mm + M5; break;
    case 6: p1 = mm + M6 + 3 * (typ == T_PROPOSE ? 0 : (typ == T_STEAL ? 1 : 2)); break;    // propose row, steal row or the dummy row
    case 9:
        p1 = mm + M9 + 5 * (typ == T_EXCHANGE ? 0 : 1);
        if (typ == T_PLAYDEV) p2 = mm + M9 + 5 * (card == C_MONO ? 2 : (card == C_YOP ? 3 : 1));
        break;
    case 10: p1 = mm + M10; break;
    case 11: p1 = mm + M11; break;
    default: {
        const bool from_hand = h == 7;
        float tot = 0.0f;
        for (int k = 0; k < 6; k++) tot += st[HS_RES + k];
#pragma unroll
        for (int q = 0; q < 20; q++) {
            const int col = c0 + q;
            mk[q] = col >= 6 ? 0.0f : (col == 0 ? ((a.step == 0 && tot != 0.0f) ? 0.0f : 1.0f) : ((from_hand && !(st[HS_RES + (col < 6 ? col : 0)] > 0.0f)) ? 0.0f : 1.0f));
        }
        return;
    }
    }
#pragma unroll
    for (int q = 0; q < 20; q++) {
        const int col = c0 + q;
        mk[q] = col < a.K ? p1[col] * (p2 != nullptr ? p2[col] : 1.0f) : 0.0f;
    }
}
// after the row's action `act` with log-prob `lp` is known: the action column(s), the joint log-prob, the state for the next heads
DEVI void hd_commit(const HeadArgs& a, long row, float* st, int act, float lp, float count) {
    long long* out = a.actions + row * 18;
    const int h = a.head_id;
    auto add = [&](float v) { st[HS_TOTAL] += count != 0.0f ? v * count : 0.0f; };
    switch (h) {
    case 0: {
        int typ = act; float l = lp;
        if (a.forced != nullptr && a.forced[row] >= 0) { typ = (int)a.forced[row]; l = 0.0f; }
        st[HS_TYP] = (float)typ; st[HS_TOTAL] = l; out[0] = typ;
        return;
    }
    case 1: out[1] = act; add(lp); return;
    case 2: out[2] = act; add(lp); return;
    case 3: out[3] = act; add(lp); return;
    case 4: out[4] = act; st[HS_CARD] = (float)act; add(lp); return;
    case 5: out[5] = act; add(lp); return;
    case 6: out[6] = act; add(lp); return;
    case 9: out[15] = act; st[HS_RA] = (float)act; st[HS_CNT9] = count; add(lp); return;
    case 10: out[16] = act; add(lp); return;
    case 11: out[17] = act; add(lp); return;
    default: {
        out[(h == 7 ? 7 : 11) + a.step] = act;
        st[HS_OUT + act] += 1.0f;
        st[HS_RES + act] = fmaxf(st[HS_RES + act] - 1.0f, 0.0f);
        const float keep = a.step == 0 ? 1.0f : (st[HS_PREV] > 0.0f ? 1.0f : 0.0f);       // a list ends at its first 0 ("stop")
        st[HS_LPSUM] += keep != 0.0f ? lp : 0.0f;
        st[HS_PREV] = (float)act;
        st[HS_OUT] = 0.0f;                                                              // column 0 never feeds back
        if (a.step == 3) {
            const float prop = (int)st[HS_TYP] == T_PROPOSE ? 1.0f : 0.0f;
            const float l = prop != 0.0f ? st[HS_LPSUM] : 0.0f;
            st[HS_TOTAL] += l;
            if (h == 7) {
                st[HS_FILT7] = l == 0.0f ? 1.0f : 0.0f;                                 // action_heads_module.py:175
                for (int k = 0; k < 6; k++) st[HS_GIVE + k] = st[HS_OUT + k];
            } else {
                a.logp_out[row] = st[HS_TOTAL];
            }
        }
        return;
    }
    }
}


// the kernels' LDS: one head's weights, a tile's logits, the rows' conditioning columns and chained state
template <int RT, int WAVES>
struct HeadShared {
    __attribute__((aligned(16))) unsigned short sW2[128 * HD_PITCH];
    __attribute__((aligned(16))) unsigned short sW3[5 * 16 * HD_PITCH];
    __attribute__((aligned(16))) unsigned short sW1[HD_NCP * 128];
    float sV[HD_VELEMS];
    __attribute__((aligned(16))) unsigned short sLg[WAVES][16 * HD_LG];   // a tile's logits (bf16, as the unfused path rounds them)
    float sCond[WAVES][RT][16][HD_NCP];     // chained mode: the rows' conditioning columns and log-prob factors
    float sCnt[WAVES][RT][16];
    __attribute__((aligned(16))) float sState[WAVES][RT][16][HD_STATE];   // the rows' chained state (LDS: ordered within the wave)
};
// One head evaluation of a workgroup's 256 rows.
template <int KT, int RT, int WAVES>
DEVI void hd_eval(const HeadArgs& a, HeadShared<RT, WAVES>& sh) {
    constexpr int HD_RT = RT, HD_WAVES = WAVES, HD_THREADS = WAVES * 64, HD_ROWS = WAVES * RT * 16;
    auto& sW2 = sh.sW2; auto& sW3 = sh.sW3; auto& sW1 = sh.sW1; auto& sV = sh.sV; auto& sLg = sh.sLg; auto& sCond = sh.sCond; auto& sCnt = sh.sCnt;
    auto& sState = sh.sState;
    const bool chained = a.state != nullptr;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, g = lane >> 4;
    const int rr = lane >> 2, part = lane & 3, c0 = part * 20;   // the categorical's split: four lanes per row, 20 columns each
    // Everything a wave's row tiles need from HBM is requested before the weights are staged: the trunk products, the rows' state,
    // and - once the state is there - the mask entries (whose address depends on the type the earlier heads chose).  There are two
    // waves per SIMD: what is left exposed is one round trip for the state and one for the masks per launch.
    uint4 xr[HD_RT][4];
    float4 sr[HD_RT][2];
    float thr[HD_RT];
#pragma unroll
    for (int tt = 0; tt < HD_RT; tt++) {
        const long row0 = (long)blockIdx.x * HD_ROWS + (wave * HD_RT + tt) * 16;
        const long row = row0 + lr < a.B ? row0 + lr : a.B - 1;        // rows past the end repeat the last one; nothing is stored for them
        const long grow = row0 + rr < a.B ? row0 + rr : a.B - 1;
#pragma unroll
        for (int s = 0; s < 4; s++) xr[tt][s] = *reinterpret_cast<const uint4*>(a.pre + row * a.pre_ld + s * 32 + g * 8);
        if (chained) {
            const float4* src = reinterpret_cast<const float4*>(a.state + grow * HD_STATE + part * 8);
            sr[tt][0] = src[0]; sr[tt][1] = src[1];
        }
        thr[tt] = a.u ? a.u[grow] : 0.0f;
    }
    for (int i = tid; i < 128 * 16; i += HD_THREADS) {
        const int n = i >> 4, c = i & 15;
        *reinterpret_cast<uint4*>(sW2 + n * HD_PITCH + c * 8) = *reinterpret_cast<const uint4*>(a.wts + n * 128 + c * 8);
    }
    for (int i = tid; i < KT * 16 * 16; i += HD_THREADS) {
        const int n = i >> 4, c = i & 15;
        *reinterpret_cast<uint4*>(sW3 + n * HD_PITCH + c * 8) = *reinterpret_cast<const uint4*>(a.wts + 128 * 128 + n * 128 + c * 8);
    }
    if (a.ncond > 0)
        for (int i = tid; i < HD_NCP * 16; i += HD_THREADS)
            *reinterpret_cast<uint4*>(sW1 + i * 8) = *reinterpret_cast<const uint4*>(a.wts + 128 * 128 + HD_KP * 128 + i * 8);
    for (int i = tid; i < HD_VELEMS; i += HD_THREADS) sV[i] = a.vec[i];
    const float* lnw = sV; const float* lnb = sV + 128; const float* b2 = sV + 256; const float* b3 = sV + 384;
    unsigned short* lg = sLg[wave];
    u32 mkb[HD_RT];                                                  // the rows' mask entries of this lane's 20 columns, as bits
    if (chained) {
#pragma unroll
        for (int tt = 0; tt < HD_RT; tt++) {
            float4* dst = reinterpret_cast<float4*>(&sState[wave][tt][rr][part * 8]);
            dst[0] = sr[tt][0]; dst[1] = sr[tt][1];
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 16 * HD_RT) {                                       // one lane per row: conditioning columns, log-prob factor
            const int tt = lane >> 4, r16 = lane & 15;
            const long row0 = (long)blockIdx.x * HD_ROWS + (wave * HD_RT + tt) * 16;
            const long r = row0 + r16 < a.B ? row0 + r16 : a.B - 1;
            sCnt[wave][tt][r16] = hd_glue(a, r, sState[wave][tt][r16], sCond[wave][tt][r16]);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int tt = 0; tt < HD_RT; tt++) {
            const long row0 = (long)blockIdx.x * HD_ROWS + (wave * HD_RT + tt) * 16;
            const long grow = row0 + rr < a.B ? row0 + rr : a.B - 1;
            float mk[20];
            hd_mask20(a, grow, sState[wave][tt][rr], c0, mk);
            u32 b = 0;
#pragma unroll
            for (int q = 0; q < 20; q++) b |= (mk[q] > 0.0f && c0 + q < KT * 16) ? 1u << q : 0u;
            mkb[tt] = b;
        }
    } else {
#pragma unroll
        for (int tt = 0; tt < HD_RT; tt++) {
            const long row0 = (long)blockIdx.x * HD_ROWS + (wave * HD_RT + tt) * 16;
            const long grow = row0 + rr < a.B ? row0 + rr : a.B - 1;
            const float* mrow = a.mask + grow * a.mask_ld;
            float mk[20];
#pragma unroll
            for (int q = 0; q < 20; q++) mk[q] = c0 + q < a.K ? mrow[c0 + q] : 0.0f;
            u32 b = 0;
#pragma unroll
            for (int q = 0; q < 20; q++) b |= (mk[q] > 0.0f && c0 + q < KT * 16) ? 1u << q : 0u;
            mkb[tt] = b;
        }
    }
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < HD_RT; tt++) {
        const long row0 = (long)blockIdx.x * HD_ROWS + (wave * HD_RT + tt) * 16;
        if (row0 >= a.B) break;                                       // (wave-uniform)
        const long row = row0 + lr < a.B ? row0 + lr : a.B - 1;
        const long grow = row0 + rr < a.B ? row0 + rr : a.B - 1;
        // ---- x = pre (+ cond . W1e^T), in the operand layout
        float x[4][8];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const uint4 u4 = xr[tt][s];
            x[s][0] = __uint_as_float(u4.x << 16); x[s][1] = __uint_as_float(u4.x & 0xFFFF0000u);
            x[s][2] = __uint_as_float(u4.y << 16); x[s][3] = __uint_as_float(u4.y & 0xFFFF0000u);
            x[s][4] = __uint_as_float(u4.z << 16); x[s][5] = __uint_as_float(u4.z & 0xFFFF0000u);
            x[s][6] = __uint_as_float(u4.w << 16); x[s][7] = __uint_as_float(u4.w & 0xFFFF0000u);
        }
        if (a.ncond > 0) {
            float acc[4][8];
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int e = 0; e < 8; e++) acc[s][e] = 0.0f;
            // four conditioning columns per round, their loads issued together (a load per column inside the loop is an exposed
            // HBM round trip each: 12 columns x 4 tiles of them made this kernel three times longer)
            const float* crow = chained ? sCond[wave][tt][lr] : a.cond + row * a.cond_ld;
            for (int j0 = 0; j0 < a.ncond; j0 += 4) {
                float cj[4];
#pragma unroll
                for (int jj = 0; jj < 4; jj++) cj[jj] = j0 + jj < a.ncond ? crow[j0 + jj] : 0.0f;
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const float cv = hd_bf(cj[jj]);
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        float wv[8];
                        te_load8(sW1 + (j0 + jj) * 128 + s * 32 + g * 8, wv);       // (rows >= ncond of the pack are zero)
#pragma unroll
                        for (int e = 0; e < 8; e++) acc[s][e] += cv * wv[e];
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int e = 0; e < 8; e++) x[s][e] = hd_bf(x[s][e] + hd_bf(acc[s][e]));
        }
        // ---- LayerNorm(128) + ReLU: the row's values sit in the lanes lr, lr + 16, lr + 32, lr + 48
        float sum = 0.0f;
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int e = 0; e < 8; e++) sum += x[s][e];
        sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / 128.0f);
        float sq = 0.0f;
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int e = 0; e < 8; e++) { const float d = x[s][e] - mean; sq += d * d; }
        sq += __shfl_xor(sq, 16); sq += __shfl_xor(sq, 32);
        const float rstd = rsqrtf(sq * (1.0f / 128.0f) + a.eps);
        bf16x8_t xa[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            unsigned short hv[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int col = s * 32 + g * 8 + e;
                hv[e] = te_to_bf(fmaxf((x[s][e] - mean) * rstd * lnw[col] + lnb[col], 0.0f));
            }
            uint4 u4;
            u4.x = (unsigned)hv[0] | ((unsigned)hv[1] << 16); u4.y = (unsigned)hv[2] | ((unsigned)hv[3] << 16);
            u4.z = (unsigned)hv[4] | ((unsigned)hv[5] << 16); u4.w = (unsigned)hv[6] | ((unsigned)hv[7] << 16);
            xa[s] = *reinterpret_cast<const bf16x8_t*>(&u4);
        }
        // ---- h^T = W2 . y^T + b2: column tile j of h lands as h[row lr][16 j + 4 g + i] in c[i]
        unsigned hp[4][4];                                            // A operand of the next product: k-step s = tiles 2 s, 2 s + 1
#pragma unroll
        for (int j = 0; j < 8; j++) {
            f32x4_t c = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const bf16x8_t wf = *reinterpret_cast<const bf16x8_t*>(sW2 + (j * 16 + lr) * HD_PITCH + s * 32 + g * 8);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xa[s], c, 0, 0, 0);
            }
            const int n0 = 16 * j + 4 * g;
            hp[j >> 1][(j & 1) * 2] = pk_bf(c[0] + b2[n0], c[1] + b2[n0 + 1]);
            hp[j >> 1][(j & 1) * 2 + 1] = pk_bf(c[2] + b2[n0 + 2], c[3] + b2[n0 + 3]);
        }
        // ---- logits = h . W3^T + b3, the contraction index in the order the lanes hold h: k = 32 s + {4 g .. 4 g + 3, 16 + 4 g .. 16 + 4 g + 3}
#pragma unroll
        for (int t = 0; t < KT; t++) {
            f32x4_t c = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const unsigned short* wr = sW3 + (t * 16 + lr) * HD_PITCH + 32 * s + 4 * g;
                const uint2 lo = *reinterpret_cast<const uint2*>(wr), hi = *reinterpret_cast<const uint2*>(wr + 16);
                uint4 bu; bu.x = lo.x; bu.y = lo.y; bu.z = hi.x; bu.w = hi.y;
                uint4 au; au.x = hp[s][0]; au.y = hp[s][1]; au.z = hp[s][2]; au.w = hp[s][3];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(&au), *reinterpret_cast<const bf16x8_t*>(&bu), c, 0, 0, 0);
            }
            const float bb = b3[16 * t + lr];
#pragma unroll
            for (int i = 0; i < 4; i++) lg[(4 * g + i) * HD_LG + 16 * t + lr] = te_to_bf(c[i] + bb);       // logits[row 4 g + i][16 t + lr]
        }
        __builtin_amdgcn_wave_barrier();
        // ---- masked categorical: four lanes per row, 20 columns each
        {
            float z[20];
            const u32 valid = mkb[tt];
            float mx = -INFINITY;
            int amax = 0x7fff;
            const unsigned* lrow = reinterpret_cast<const unsigned*>(lg + rr * HD_LG + c0);
#pragma unroll
            for (int q = 0; q < 20; q += 2) {
                const unsigned pr = lrow[q >> 1];
                z[q] = __uint_as_float(pr << 16); z[q + 1] = __uint_as_float(pr & 0xFFFF0000u);
            }
#pragma unroll
            for (int q = 0; q < 20; q++) {
                if (((valid >> q) & 1u) && z[q] > mx) { mx = z[q]; amax = c0 + q; }
            }
#pragma unroll
            for (int d = 1; d <= 2; d <<= 1) {
                const float om = __shfl_xor(mx, d); const int oa = __shfl_xor(amax, d);
                if (om > mx || (om == mx && oa < amax)) { mx = om; amax = oa; }
            }
            if (amax == 0x7fff) amax = 0;
            float sum2 = 0.0f;
#pragma unroll
            for (int q = 0; q < 20; q++) if ((valid >> q) & 1u) sum2 += __expf(z[q] - mx);
            float tot = sum2;
            tot += __shfl_xor(tot, 1); tot += __shfl_xor(tot, 2);
            const float lse = mx + __logf(tot);
            // cumulative probabilities in column order: exclusive prefix over the four parts, then in-lane
            float mine = 0.0f;
#pragma unroll
            for (int q = 0; q < 20; q++) if ((valid >> q) & 1u) mine += __expf(z[q] - lse);
            float incl = mine;
            { const float v1 = __shfl_up(incl, 1); if (part >= 1) incl += v1; }
            { const float v2 = __shfl_up(incl, 2); if (part >= 2) incl += v2; }
            float cdf = incl - mine;
            int pick = 0x7fff, last = -1;
#pragma unroll
            for (int q = 0; q < 20; q++) if ((valid >> q) & 1u) {
                cdf += __expf(z[q] - lse);
                last = c0 + q;
                if (pick == 0x7fff && cdf > thr[tt]) pick = c0 + q;
            }
#pragma unroll
            for (int d = 1; d <= 2; d <<= 1) {
                pick = min(pick, __shfl_xor(pick, d));
                last = max(last, __shfl_xor(last, d));
            }
            int act = a.u ? (pick != 0x7fff ? pick : (last >= 0 ? last : amax)) : amax;
            act = min(max(act, 0), a.K - 1);
            // is the chosen column legal?  its mask bit sits in the lane of this row that holds column `act`
            const u32 vb = __shfl(valid, (lane & ~3) | (act / 20));
            if (part == 0 && row0 + rr < a.B) {
                const float lpa = (((vb >> (act % 20)) & 1u) ? te_bf(lg[rr * HD_LG + act]) : -INFINITY) - lse;
                if (chained) hd_commit(a, grow, sState[wave][tt][rr], act, lpa, sCnt[wave][tt][rr]);
                else { a.action[grow] = act; a.logp[grow] = lpa; }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (chained && row0 + rr < a.B) {                      // the rows' state goes back for the next evaluation
            float4* dst = reinterpret_cast<float4*>(a.state + (row0 + rr) * HD_STATE + part * 8);
            const float4* src = reinterpret_cast<const float4*>(&sState[wave][tt][rr][part * 8]);
            dst[0] = src[0]; dst[1] = src[1];
        }
    }
}

template <int KT, int RT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_head_fwd(HeadArgs a) {
    __shared__ HeadShared<RT, WAVES> sh;
    hd_eval<KT, RT, WAVES>(a, sh);
}
}  // namespace catan
