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

// ------------------------------------------------------------------------------------------------ k_obs_rows (round 3)
// The same observation, built for the memory system.  k_obs above spends its time issuing instructions: lane = game fills
// 1 787 floats one LDS store at a time and the wave then writes every game's chunks with 4-byte lanes (~2 500 store
// instructions of <= 256 B per wave; 415 us for 65 536 games = 16 % of the HBM peak on its own bytes).  Here:
//   * every observation value is a small multiple of 1/8 (one-hots, counts, n/8, n/4), so a value is ONE BYTE in LDS
//     (code = 8 x value <= 255, static bounds below); the tile of a wave's 16 games is 28 KB and is zero-filled with 16-byte
//     LDS stores - the generators only write the non-zero entries;
//   * four lanes per game: lane q of a game's quad does the tile corners 2q-2, 2q-1 (q = 0: the tile headers) of all 19 tiles,
//     one of the four player vectors and one or two of the card lists;
//   * the write-out converts 8 (bf16) or 4 (fp32) codes per lane to one 16-byte store: the dense [n][1787] matrix the policy
//     reads is ONE contiguous, 16-byte aligned block per wave; the rollout-storage row of a game (obs_f[t[g]][g][:], selected by
//     sel[g]) starts at any element, so its first / last few elements go out element-wise and the rest in aligned chunks read
//     from LDS through a funnel shift.  bf16 is exact for every code (8 significant bits).
// One pass writes what k_obs + the fp32 -> bf16 cast + catan_masked_row_store wrote in three.
constexpr int OBS_OG = 8;                     // games per wave (16 / 8 / 4: four / eight / sixteen lanes per game; fewer games = less LDS, more waves per SIMD)
static_assert(OBS_FLOATS == 1787 && (8 * OBS_FLOATS * 2) % 16 == 0 && (8 * OBS_FLOATS * 4) % 16 == 0, "a wave's dense block is 16-byte aligned");

struct ObsF32 { typedef float T; static constexpr int EPC = 4; };
struct ObsBF16 { typedef unsigned short T; static constexpr int EPC = 8; };
DEVI float obs_val(u32 code) { return (float)code * 0.125f; }
DEVI u32 obs_bf16(u32 code) { return __float_as_uint((float)code * 0.125f) >> 16; }       // exact: code < 256
DEVI void obs_store_elem(float* dst, int e, u32 code) { dst[e] = obs_val(code); }
DEVI void obs_store_elem(unsigned short* dst, int e, u32 code) { dst[e] = (unsigned short)obs_bf16(code); }
// EPC codes starting at byte `e0` of the LDS code array (any alignment; the array is readable 16 bytes past its end)
// two codes (bytes B0, B1 of w) -> two bf16 in one dword: v_cvt_f32_ubyteN, x 1/8, and one byte permute that keeps the high halves
template <int B0, int B1>
DEVI u32 obs_pair(u32 w) {
    const float a = (float)((w >> (8 * B0)) & 255u) * 0.125f, b = (float)((w >> (8 * B1)) & 255u) * 0.125f;
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);      // { a.hi16, b.hi16 }
}
DEVI uint4 obs_chunk(const u8* codes, int e0, ObsBF16) {
    const u32* p = reinterpret_cast<const u32*>(codes + (e0 & ~3));
    const u32 w0 = p[0], w1 = p[1], w2 = p[2];
    const u32 sh = (u32)(e0 & 3) * 8u;
    const u32 lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);   // ({w1,w0} >> sh) & 0xffffffff
    uint4 o;
    o.x = obs_pair<0, 1>(lo); o.y = obs_pair<2, 3>(lo); o.z = obs_pair<0, 1>(hi); o.w = obs_pair<2, 3>(hi);
    return o;
}
DEVI uint4 obs_chunk(const u8* codes, int e0, ObsF32) {
    const u32* p = reinterpret_cast<const u32*>(codes + (e0 & ~3));
    const u32 w = __builtin_amdgcn_alignbit(p[1], p[0], (u32)(e0 & 3) * 8u);
    uint4 o;
    o.x = __float_as_uint(obs_val(w & 255u)); o.y = __float_as_uint(obs_val((w >> 8) & 255u));
    o.z = __float_as_uint(obs_val((w >> 16) & 255u)); o.w = __float_as_uint(obs_val(w >> 24));
    return o;
}
// `count` elements codes[first ..] -> dst[0 ..] (dst aligned to its element size only): the elements up to the first 16-byte
// boundary and behind the last one go out one per lane, the rest as aligned 16-byte stores.  Whole wave, uniform arguments.
template <class O>
DEVI void obs_write_span(const u8* codes, int first, typename O::T* dst, int count, int lane) {
    constexpr int EPC = O::EPC, ES = (int)sizeof(typename O::T);
    const int head = min(count, (int)(((16u - (u32)((uintptr_t)dst & 15u)) & 15u) / ES));
    const int nfull = (count - head) / EPC;
    const int tail0 = head + nfull * EPC, ntail = count - tail0;
    if (lane < head) obs_store_elem(dst, lane, codes[first + lane]);
    else if (lane - head < ntail) obs_store_elem(dst, tail0 + lane - head, codes[first + tail0 + lane - head]);
    uint4* d4 = reinterpret_cast<uint4*>(dst + head);
    for (int ch = lane; ch < nfull; ch += 64) d4[ch] = obs_chunk(codes, first + head + ch * EPC, O());
}

// player vector of `target` (152 codes for label < 0, else 159) into row[]; env/wrapper.py:526-709.  Only non-zero entries.
template <class S>
DEVI void obs_player_codes(const S& s, u8* codes0, u8* row, int me, int label, int order, int seatof, const u64 (&bld)[4], const u64 (&cty)[4]) {
    const int ro[5] = { R_WOOD, R_BRICK, R_WHEAT, R_ORE, R_SHEEP };                            // wrapper.py:550
    const int target = label < 0 ? me : player_at_label(order, seatof, me, label);
    int k = 0;
    if (label < 0) {
#pragma unroll
        for (int i = 0; i < 5; i++) { row[k + bucket8(s.res(target, ro[i]))] = 8; k += 8; }
    } else {
        Est E;
        est_load(s, me, label, E);
#pragma unroll
        for (int i = 0; i < 5; i++) { row[k + bucket8(E.mn[ro[i]])] = 8; k += 8; }
#pragma unroll
        for (int i = 0; i < 5; i++) { row[k + bucket8(E.mx[ro[i]])] = 8; k += 8; }
    }
    const int vp = s.pb(target, P_VP);
    row[k + (vp < 10 ? vp : 9)] = 8; k += 10;                                                  // :588-593
    {                                                                                          // :596-610
        // index of resource r0 in `ro` (Wood, Brick, Wheat, Ore, Sheep), four bits each: Brick 1, Wood 0, Ore 3, Sheep 4, Wheat 2
        constexpr u32 RI = (1u << (4 * R_BRICK)) | (0u << (4 * R_WOOD)) | (3u << (4 * R_ORE)) | (4u << (4 * R_SHEEP)) | (2u << (4 * R_WHEAT));
        const u64 tb_ = bld[target], tc_ = cty[target];
#pragma unroll
        for (int t = 0; t < 19; t++) {
            const int tb = s.b(B_TILE + t), v = tb >> 4, r0 = (tb & 15) - 1;
            const u64 tm = topo_tile_corners(t);
            const int cnt = __popcll(tb_ & tm) + 2 * __popcll(tc_ & tm);
            if (v == 7 || cnt == 0) continue;
            const int ri = (int)((RI >> (4 * r0)) & 15u);
            const int ind = v <= 6 ? v - 2 : v - 3;
            // += 8 cnt on the byte: an LDS add on the containing word (no carry: a cell collects <= 2 tiles x 3 buildings x 2 =
            // 12 -> code 96), fire and forget instead of a read-modify-write chain through the LDS latency
            const u32 a = (u32)(row - codes0) + (u32)(k + ri * 10 + ind);
            atomicAdd(reinterpret_cast<u32*>(codes0) + (a >> 2), (u32)(8 * cnt) << (8 * (a & 3u)));
        }
    }
    k += 50;
    const int lrp = s.b(B_LR_PLAYER);
    if (lrp) {                                                                                 // :613-620 (value n/8 -> code n)
        if (lrp == target + 1) { row[k] = 8; row[k + 1] = (u8)s.b(B_LR_COUNT); }
        else row[k + 1] = (u8)s.pb(target, P_CURLP);
    }
    k += 2;
    if (s.b(B_LA_PLAYER) == target + 1) row[k] = 8;                                            // :623-627 (n/4 -> code 2n)
    row[k + 1] = (u8)(2 * s.pb(target, P_ARMY));
    k += 2;
    const int hb = s.pb(target, P_HARB);
#pragma unroll
    for (int i = 0; i < 6; i++) if ((hb >> i) & 1) row[k + i] = 8;                             // :632-637
    k += 6;
    if (label < 0) {
#pragma unroll
        for (int i = 0; i < 5; i++) { row[k + bucket7(s.b(B_BANK + ro[i]))] = 8; k += 7; }    // :657-672
        row[k + bucket7(s.b(B_PILE_LEN))] = 8; k += 7;                                         // :674-686
    } else {
        row[k + label] = 8; k += 3;                                                            // :532-541
        const int nh = s.pb(target, P_NHID);
        row[k + (nh <= 4 ? nh : 5)] = 8; k += 6;                                               // :690-695
    }
}

// dense_f / dense_lists / dense_lens: [n][1787] / int32 [n][5][25] / int32 [n][5] for all games (any of the three groups may be
// null).  rows_f / rows_lists / rows_lens: the rollout storage ([steps][n][1787] of the same element type, int8 [steps][n][5][25],
// int8 [steps][n][5]); game g with sel[g] != 0 writes its row at step t_idx[g] (rows_* null: no row stores).
// OG games per wave, LPG = 64 / OG lanes per game.  Roles of lane r of a game's group:
//   LPG = 4: r = 0 tile headers + trade / resources, r = 1..3 tile corners 2r-2, 2r-1; player vector r; card list r (r = 3: also list 4)
//   LPG = 8: r = 0 tile headers + trade / resources, r = 1..6 tile corner r-1; player vector r-4 on r = 4..7; card list r on r = 0..4
//   LPG = 16: r = 0 tile headers + trade / resources, r = 1..6 tile corner r-1; player vector r-7 on r = 7..10; card list r-11 on r = 11..15
// view of a whole record staged linearly in LDS (read-only use)
struct StLF : StOps<StLF> {
    const u32* T;
    DEVI explicit StLF(const u32* T_) : T(T_) {}
    DEVI u32 w(int r) const { return T[r]; }
    DEVI int b(int f) const { return ((const u8*)(T + NW))[f]; }
    DEVI int cold(int f) const { return ((const u8*)(T + NW))[f]; }
    DEVI void sw(int, u32) const {}
    DEVI void sb(int, int) const {}
    DEVI void scold(int, int) const {}
};

template <class O, int OG>
__global__ __launch_bounds__(64) void k_obs_rows(Ctx c, typename O::T* __restrict__ dense_f, i32* __restrict__ dense_lists, i32* __restrict__ dense_lens,
                                                  typename O::T* __restrict__ rows_f, signed char* __restrict__ rows_lists,
                                                  signed char* __restrict__ rows_lens, const long long* __restrict__ t_idx,
                                                  const u8* __restrict__ sel, const i32* __restrict__ games, long n_rows) {
    constexpr int LPG = 64 / OG, TILE_B = OG * OBS_FLOATS;
    static_assert(LPG == 4 || LPG == 8 || LPG == 16, "four, eight or sixteen lanes per game");
    __shared__ __attribute__((aligned(16))) u8 codes[TILE_B + 32];
    __shared__ u8 lst[OG][5][25];
    __shared__ u8 lln[OG][5];
    __shared__ u8 tcn[19 * 6];
    __shared__ __attribute__((aligned(16))) u32 rec[OG * REC];          // the records of the wave's games (hot part + ordered card lists), linear
    const int lane = threadIdx.x, gi = lane / LPG, r = lane % LPG;
    // dense row g0 + j is the observation of game gm(j): the row number itself, or games[g0 + j] (catan_obs_rows_of: a caller
    // that evaluates only some of the games); a negative / out-of-range id is an empty slot
    const long g0 = (long)blockIdx.x * OG;
    long my_game = -1;                                      // lane j < OG: the game of slot j
    if (lane < OG && g0 + lane < n_rows) { my_game = games != nullptr ? (long)games[g0 + lane] : g0 + lane; if (my_game >= c.n) my_game = -1; }
    int my_sel = 0; long long my_t = 0;                     // ... does it append its row, and at which step
    if (rows_f != nullptr && my_game >= 0) { my_sel = sel[my_game]; my_t = t_idx[my_game]; }
    {   // the records (704 B each) with 16-byte loads, all in flight at once; meanwhile zero the code tile (16-byte LDS
        // stores) and copy the tile-corner table to LDS (per-lane indices below)
        constexpr int CH = REC / 4, NV = (OG * CH + 63) / 64;                 // 44 chunks per record
        uint4 v[NV];
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int k = i * 64 + lane, gg = k / CH, ch = k - gg * CH;
            long ge = __shfl(my_game, gg < OG ? gg : 0);
            if (ge < 0) ge = c.N - 1;                                          // empty slots read a valid record (padding games are)
            v[i] = make_uint4(0, 0, 0, 0);
            if (k < OG * CH) v[i] = reinterpret_cast<const uint4*>(c.R + ge * REC)[ch];
        }
        uint4* z = reinterpret_cast<uint4*>(codes);
        for (int i = lane; i < (TILE_B + 32) / 16; i += 64) z[i] = make_uint4(0, 0, 0, 0);
        for (int i = lane; i < 19 * 6; i += 64) tcn[i] = TILE_CORNER[i / 6][i % 6];
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int k = i * 64 + lane;
            if (k < OG * CH) reinterpret_cast<uint4*>(rec)[k] = v[i];
        }
    }
    __syncthreads();
    StLF s(rec + gi * REC);
    u8* row = codes + gi * OBS_FLOATS;
    const int order = s.b(B_ORDER), seatof = s.b(B_SEATOF);
    const int flags = s.flags();
    int me;                                                                                    // wrapper.py:53-58
    if (s.b(B_NDISC) > 0) me = s.b(B_DISC);
    else if (flags & F_MUST_RESPOND) me = s.b(B_TRADE_TGT);
    else me = s.b(B_GO);
    u64 bld[4], cty[4];
#pragma unroll
    for (int p = 0; p < 4; p++) { bld[p] = s.settle(p); cty[p] = s.city(p); }
    // ---- tile_representations (wrapper.py:491-524)
    if (r == 0) {
        const int robber = s.b(B_ROBBER);
#pragma unroll
        for (int t = 0; t < 19; t++) {
            u8* f = row + OBS_OFF_TILES + t * 60;
            const int tb = s.b(B_TILE + t);
            if (t == robber) f[0] = 8;
            f[1 + (tb >> 4) - 2] = 8;
            f[12 + (tb & 15)] = 8;
        }
        // proposed_trade + current_resources (wrapper.py:60-71)
        if (flags & F_MUST_RESPOND) {
            const int ng = s.b(B_TRADE_NG), nr = s.b(B_TRADE_NR);
            for (int i = 0; i < 4; i++) if (i < ng) row[s.b(B_TRADE_GIVE + i)] = 8;
            for (int i = 0; i < 4; i++) if (i < nr) row[s.b(B_TRADE_RECV + i) + 5] = 8;
        }
        for (int rr = 0; rr < 5; rr++) row[12 + rr + 1] = (u8)(8 * s.res(me, rr));               // <= 19 cards of a resource exist
    } else if (r <= (LPG == 4 ? 3 : 6)) {
        constexpr int KPL = LPG == 4 ? 2 : 1;                                                 // corners per lane
        // building / relative owner of the lane's corners: owner code 0 self, 1 + label otherwise
        u32 relp = 0;                                                                          // rel[p] in bits 4p..4p+3
#pragma unroll
        for (int p = 0; p < 4; p++) relp |= (u32)(p == me ? 0 : 1 + label_of(seatof, me, p)) << (4 * p);
        u64 any_b = bld[0] | bld[1] | bld[2] | bld[3], any_c = cty[0] | cty[1] | cty[2] | cty[3];
#pragma unroll
        for (int t = 0; t < 19; t++) {
#pragma unroll
            for (int kk = 0; kk < KPL; kk++) {
                const int k = KPL * (r - 1) + kk;
                const int cn = tcn[t * 6 + k];
                u8* gq = row + OBS_OFF_TILES + t * 60 + 18 + k * 7;
                const int b = ((any_c >> cn) & 1) ? 2 : (int)((any_b >> cn) & 1);
                gq[b] = 8;
                if (b) {
                    int ow = 0;
#pragma unroll
                    for (int p = 1; p < 4; p++) if (((bld[p] | cty[p]) >> cn) & 1) ow = p;
                    gq[3 + ((relp >> (4 * ow)) & 15)] = 8;
                }
            }
        }
    }
    // ---- player vectors: the deciding player's (label -1) and the three opponents' (labels 0..2)
    {
        const int j = LPG == 4 ? r : (LPG == 8 ? r - 4 : r - 7);                               // which vector this lane writes
        if (j >= 0 && j < 4) obs_player_codes(s, codes, row + (j == 0 ? OBS_OFF_CUR : OBS_OFF_OTHER + (j - 1) * 159), me, j - 1, order, seatof, bld, cty);
    }
    // ---- dev-card id lists (wrapper.py:642-655): ids = card + 1, [0] when empty
    for (int li = (LPG == 16 ? r - 11 : r); li >= 0 && li < 5; li += (LPG == 4 && r == 3) ? 1 : 8) {
        const int who = li < 2 ? me : player_at_label(order, seatof, me, li - 2);
        const bool hid = li == 1;
        const int cnt = hid ? s.pb(who, P_NHID) : s.pb(who, P_NPLAYED);
        for (int i = 0; i < 25; i++) lst[gi][li][i] = (u8)(i < cnt ? (hid ? s.hidden(who, i) : s.played(who, i)) + 1 : 0);
        lln[gi][li] = (u8)(cnt == 0 ? 1 : cnt);
    }
    __syncthreads();
    // ---- write-out
    const int ng = (int)min((long)OG, n_rows - g0);          // rows of this wave
    if (ng <= 0) return;
    if (dense_f != nullptr) obs_write_span<O>(codes, 0, dense_f + g0 * OBS_FLOATS, ng * OBS_FLOATS, lane);
    if (dense_lists != nullptr) {
        const u8* l8 = &lst[0][0][0];
        for (int i = lane; i < ng * 125; i += 64) dense_lists[g0 * 125 + i] = l8[i];
        const u8* n8 = &lln[0][0];
        for (int i = lane; i < ng * 5; i += 64) dense_lens[g0 * 5 + i] = n8[i];
    }
    if (rows_f != nullptr) {
        for (int j = 0; j < ng; j++) {
            const long gj = __shfl(my_game, j);
            if (!__shfl(my_sel, j)) continue;              // (uniform)
            const long rw = (long)__shfl(my_t, j) * c.n + gj;
            obs_write_span<O>(codes, j * OBS_FLOATS, rows_f + rw * OBS_FLOATS, OBS_FLOATS, lane);
            if (rows_lists != nullptr) {
                const u8* l8 = &lst[j][0][0];
                for (int i = lane; i < 125; i += 64) rows_lists[rw * 125 + i] = (signed char)l8[i];
                if (lane < 5) rows_lens[rw * 5 + lane] = (signed char)lln[j][lane];
            }
        }
    }
}

}  // namespace catan
