// catan_obs.hip - observation encoder (EnvWrapper._get_obs, reference env/wrapper.py:52-83, 491-709).
//
// Output per game: float32[1787] = proposed_trade[12] | current_resources[6] | tile_representations[19][60] |
// current_player_main[152] | next/next_next/next_next_next_player_main[159] each, plus the five dev-card id lists
// (current played, current hidden, next x3 played) zero-padded to 25 with their lengths ([0] when empty, len 1).
// This is the one HBM-bound kernel of the path: 7.3 KB written per game against ~0.5 KB read.
//
// One wave = 64 games.  Lane = game computes its features chunk by chunk into an LDS staging tile
// (row stride 161 words: odd, so the 64 lanes hit distinct banks); then the wave writes the rows out with every
// store covering one contiguous <= 256 B segment of a game's row (instead of 64 scattered 4 B stores).
#include <hip/hip_runtime.h>

namespace catan {

constexpr int OBS_STRIDE = 161;
constexpr int OBS_OFF_TRADE = 0, OBS_OFF_RES = 12, OBS_OFF_TILES = 18, OBS_OFF_CUR = 1158, OBS_OFF_OTHER = 1310;

DEVI int bucket8(int v) { return v < 5 ? v : (v < 8 ? 5 : (v < 11 ? 6 : 7)); }                 // wrapper.py:552-562
DEVI int bucket7(int v) { return v <= 2 ? v : (v <= 5 ? 3 : (v <= 7 ? 4 : (v <= 10 ? 5 : 6))); }   // wrapper.py:660-686

// cooperative write-out of `len` floats per game: rows[g][0..len) -> out[(g0+g) * 1787 + off + ...]
DEVI void obs_flush(const float* rows, float* __restrict__ out, long g0, long n, int off, int len, int lane) {
    __builtin_amdgcn_wave_barrier();
    for (int g = 0; g < 64; g++) {
        if (g0 + g >= n) break;
        float* dst = out + (g0 + g) * (long)OBS_FLOATS + off;
        for (int j = lane; j < len; j += 64) dst[j] = rows[g * OBS_STRIDE + j];
    }
    __builtin_amdgcn_wave_barrier();
}

// env/wrapper.py:526-709 for one target player; writes 152 (label < 0) or 159 floats into row
DEVI void obs_player(const St& s, float* row, int me, int label, int order, int seatof, const u64 (&bld)[4], const u64 (&cty)[4]) {
    const int ro[5] = { R_WOOD, R_BRICK, R_WHEAT, R_ORE, R_SHEEP };                            // wrapper.py:550
    const int target = label < 0 ? me : player_at_label(order, seatof, me, label);
    const int n = label < 0 ? 152 : 159;
    for (int i = 0; i < n; i++) row[i] = 0.0f;
    int k = 0;
    if (label < 0) {
#pragma unroll
        for (int i = 0; i < 5; i++) { row[k + bucket8(s.res(target, ro[i]))] = 1.0f; k += 8; }
    } else {
        Est E;
        est_load(s, me, label, E);
#pragma unroll
        for (int i = 0; i < 5; i++) { row[k + bucket8(E.mn[ro[i]])] = 1.0f; k += 8; }
#pragma unroll
        for (int i = 0; i < 5; i++) { row[k + bucket8(E.mx[ro[i]])] = 1.0f; k += 8; }
    }
    const int vp = s.pb(target, P_VP);
    row[k + (vp < 10 ? vp : 9)] = 1.0f; k += 10;                                               // :588-593
    // resource access histogram (:596-610): every building of the target on a numbered tile
    for (int t = 0; t < 19; t++) {
        const int tb = s.b(B_TILE + t), v = tb >> 4, r0 = (tb & 15) - 1;
        if (v == 7) continue;
        const u64 tm = topo_tile_corners(t);
        const int cnt = __popcll(bld[target] & tm) + 2 * __popcll(cty[target] & tm);
        if (cnt == 0) continue;
        int ri = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) if (ro[i] == r0) ri = i;
        const int ind = v <= 6 ? v - 2 : v - 3;
        row[k + ri * 10 + ind] += (float)cnt;
    }
    k += 50;
    const int lrp = s.b(B_LR_PLAYER);
    if (lrp) {                                                                                 // :613-620
        if (lrp == target + 1) { row[k] = 1.0f; row[k + 1] = (float)((double)s.b(B_LR_COUNT) / 8.0); }
        else row[k + 1] = (float)((double)s.pb(target, P_CURLP) / 8.0);
    }
    k += 2;
    if (s.b(B_LA_PLAYER) == target + 1) row[k] = 1.0f;                                         // :623-627
    row[k + 1] = (float)((double)s.pb(target, P_ARMY) / 4.0);
    k += 2;
    const int hb = s.pb(target, P_HARB);
#pragma unroll
    for (int i = 0; i < 6; i++) if ((hb >> i) & 1) row[k + i] = 1.0f;                          // :632-637
    k += 6;
    if (label < 0) {
#pragma unroll
        for (int i = 0; i < 5; i++) { row[k + bucket7(s.b(B_BANK + ro[i]))] = 1.0f; k += 7; } // :657-672
        row[k + bucket7(s.b(B_PILE_LEN))] = 1.0f; k += 7;                                      // :674-686
    } else {
        row[k + label] = 1.0f; k += 3;                                                         // :532-541
        const int nh = s.pb(target, P_NHID);
        row[k + (nh <= 4 ? nh : 5)] = 1.0f; k += 6;                                            // :690-695
    }
}

// out_f: float [n][1787]; out_lists: int32 [n][5][25]; out_lens: int32 [n][5]
__global__ __launch_bounds__(64) void k_obs(Ctx c, float* __restrict__ out_f, i32* __restrict__ out_lists, i32* __restrict__ out_lens) {
    __shared__ float rows[64 * OBS_STRIDE];
    const int lane = threadIdx.x;
    const long g0 = (long)blockIdx.x * 64;
    St s(c.R, c.N, g0 + lane);            // padding games are valid rows
    float* row = rows + lane * OBS_STRIDE;
    const int order = s.b(B_ORDER), seatof = s.b(B_SEATOF);
    const int flags = s.flags();
    int me;                                                                                    // wrapper.py:53-58
    if (s.b(B_NDISC) > 0) me = s.b(B_DISC);
    else if (flags & F_MUST_RESPOND) me = s.b(B_TRADE_TGT);
    else me = s.b(B_GO);
    u64 bld[4], cty[4];
#pragma unroll
    for (int p = 0; p < 4; p++) { bld[p] = s.settle(p); cty[p] = s.city(p); }
    // ---- proposed_trade + current_resources (wrapper.py:60-71)
    for (int i = 0; i < 18; i++) row[i] = 0.0f;
    if (flags & F_MUST_RESPOND) {
        const int ng = s.b(B_TRADE_NG), nr = s.b(B_TRADE_NR);
        for (int i = 0; i < 4; i++) if (i < ng) row[s.b(B_TRADE_GIVE + i)] = 1.0f;
        for (int i = 0; i < 4; i++) if (i < nr) row[s.b(B_TRADE_RECV + i) + 5] = 1.0f;
    }
    for (int r = 0; r < 5; r++) row[12 + r + 1] = (float)s.res(me, r);
    obs_flush(rows, out_f, g0, c.n, OBS_OFF_TRADE, 18, lane);
    // ---- tile_representations (wrapper.py:491-524): three tiles (180 floats) per flush would not fit the 161 stride;
    // two tiles per pass
    const int robber = s.b(B_ROBBER);
    // relative owner index of each pid0 as seen by `me`: 0 self, 1 + label otherwise
    int rel[4];
#pragma unroll
    for (int p = 0; p < 4; p++) rel[p] = p == me ? 0 : 1 + label_of(seatof, me, p);
    for (int t0 = 0; t0 < 19; t0 += 2) {
        const int nt = t0 + 1 < 19 ? 2 : 1;
        for (int q = 0; q < nt; q++) {
            const int t = t0 + q;
            float* f = row + q * 60;
            for (int i = 0; i < 60; i++) f[i] = 0.0f;
            const int tb = s.b(B_TILE + t);
            f[0] = t == robber ? 1.0f : 0.0f;
            f[1 + (tb >> 4) - 2] = 1.0f;
            f[12 + (tb & 15)] = 1.0f;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const int cn = TILE_CORNER[t][k];
                float* g = f + 18 + k * 7;
                int b = 0, ow = -1;
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    if ((bld[p] >> cn) & 1) { b = 1; ow = p; }
                    if ((cty[p] >> cn) & 1) { b = 2; ow = p; }
                }
                g[b] = 1.0f;
                if (ow >= 0) {
                    int rv = 0;
#pragma unroll
                    for (int p = 0; p < 4; p++) if (p == ow) rv = rel[p];
                    g[3 + rv] = 1.0f;
                }
            }
        }
        obs_flush(rows, out_f, g0, c.n, OBS_OFF_TILES + t0 * 60, nt * 60, lane);
    }
    // ---- player vectors
    obs_player(s, row, me, -1, order, seatof, bld, cty);
    obs_flush(rows, out_f, g0, c.n, OBS_OFF_CUR, 152, lane);
    for (int l = 0; l < 3; l++) {
        obs_player(s, row, me, l, order, seatof, bld, cty);
        obs_flush(rows, out_f, g0, c.n, OBS_OFF_OTHER + l * 159, 159, lane);
    }
    // ---- dev-card id lists (wrapper.py:642-655): ids = card + 1, [0] when empty
    if (g0 + lane < c.n) {
        i32* lp = out_lists + (g0 + lane) * 125;
        i32* ln = out_lens + (g0 + lane) * 5;
        for (int li = 0; li < 5; li++) {
            const int who = li < 2 ? me : player_at_label(order, seatof, me, li - 2);
            const bool hid = li == 1;
            const int cnt = hid ? s.pb(who, P_NHID) : s.pb(who, P_NPLAYED);
            for (int i = 0; i < 25; i++) lp[li * 25 + i] = i < cnt ? (hid ? s.hidden(who, i) : s.played(who, i)) + 1 : 0;
            ln[li] = cnt == 0 ? 1 : cnt;
        }
    }
}

}  // namespace catan
