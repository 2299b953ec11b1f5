// catan_kernels.hip - hand-written gfx950 kernels for the batched Catan env.
//
// What each kernel replaces in the reference (henrycharlesworth/settlers_of_catan_RL):
//   k_reset, k_reset_list    Board.reset + Game.reset + EnvWrapper.reset   game/components/board.py:67-100, game/game.py:39-136, env/wrapper.py:30-34
//   k_step                   EnvWrapper.step = _translate_action + Game.apply_action + _get_done_and_rewards, and the next masks
//                                                                          env/wrapper.py:36-50,114-166,85-112, game/game.py:527-815
//   k_lr_finish, k_lr_heavy                  update_longest_road / get_longest_path + the rest of those steps   game/game.py:843-919
//   k_masks                  EnvWrapper.get_action_masks                   env/wrapper.py:168-412
//   k_sample_random          uniform-random legal policy (bench config 2)  (reference: none; rule in DESIGN.md)
//   k_classify_*             counting sort of the games by action type     (enables type-homogeneous waves)
//   k_export/import          Game.save_current_state / restore_state       game/game.py:1013-1205
//   k_expand_masks           packed 325-bit masks -> float32 [n][325]      (wrapper returns float arrays)
//   k_randomise_uncertainty  Game.randomise_uncertainty                    game/game.py:1207-1282
//
// Design notes (gfx950 / wave64; measurements and history in DESIGN.md 4):
//  * game-major 704 B records (catan_state.h).  k_step: one wave = 64 games of ONE action type (the games are sorted by
//    type every pass), their hot 448 B staged transposed in LDS so that lane = game runs the step out of LDS columns.
//  * board occupancy is kept as bitboards (54-bit corners, 72-bit edges) so placement legality, production and
//    the robber/steal masks are a handful of 64-bit and/popcount ops instead of graph walks.
//  * the rare serial work gets its own kernels, one game per wave: longest road (vertex-simple longest path: lanes run
//    the DFS from different start corners and share work through an LDS queue; tier 2 = 1 024-thread workgroups) and the
//    re-deal of a finished game (64 lanes generate the Philox words, one lane runs the shuffles).
//  * integer/byte work only - no MFMA, by design (HBM/latency-bound; see DESIGN.md roofline section).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "catan_state.h"

#define CATAN_TABLE static __device__ __constant__ const
#define CATAN_FN static __device__ __forceinline__
#define CATAN_TOPOLOGY_CODE
#include "catan_topology.inc"

namespace catan {

#define DEVI __device__ __forceinline__

constexpr int BLOCK = 256;
constexpr u64 ALL54 = (1ull << 54) - 1;

// ------------------------------------------------------------------------------------------------ state view
// RNG contract (A) on the device (single-game handles, catan_seed_mt19937): the two process-global Mersenne Twisters of the
// reference - numpy's legacy RandomState behind np.random.shuffle / randint, CPython's `random` behind random.choice - as two MT19937
// generators in device memory.  Each keeps the last MT_RING of its outputs: draw number d of a stream is ring[d % MT_RING], valid from the
// consumer's position up to `produced` (k_mt_refill runs in front of every kernel that may draw and keeps MT_AHEAD outputs ahead), so the
// kernels read "the value of draw d" exactly as they compute it from the Philox counter under contract (B) - including the re-deal's
// look-ahead over the next 1 536 draws.  The numpy stream's position is the game's own draw counter (record word W_RNG; a speculative
// re-deal that is not taken does not move it), the `random` stream's position is cons_py.
constexpr int MT_RING = 8192, MT_AHEAD = 6144, MT_AHEAD_PY = 1024;
struct MtGen { u32 mt[624]; u32 idx; u32 produced; u32 ring[MT_RING]; };
struct MtPair { MtGen np, py; u32 cons_py; };
struct Ctx {          // launch-invariant handle fields
    u32* R;           // game records, REC words each
    long N;           // padded number of games (row pitch)
    long n;           // real number of games
    u32 key0, key1;   // philox key = seed
    u64 env_id0;      // global id of game 0 (multi-GPU shards keep their global ids)
    MtPair* mt;       // contract (A): the handle's two MT19937 generators (n == 1), or null = contract (B), per-game Philox streams
};

// Derived helpers shared by the two state views (CRTP).
template <class D>
struct StOps {
    DEVI const D& self() const { return *static_cast<const D*>(this); }
    DEVI int pb(int p, int f) const { return self().b(B_PLAYER + p * PB + f); }
    DEVI void spb(int p, int f, int v) const { self().sb(B_PLAYER + p * PB + f, v); }
    DEVI u64 settle(int p) const { return (u64)self().w(W_SETTLE_LO + p) | ((u64)self().w(W_SETTLE_HI + p) << 32); }
    DEVI u64 city(int p) const { return (u64)self().w(W_CITY_LO + p) | ((u64)self().w(W_CITY_HI + p) << 32); }
    DEVI void set_settle(int p, u64 v) const { self().sw(W_SETTLE_LO + p, (u32)v); self().sw(W_SETTLE_HI + p, (u32)(v >> 32)); }
    DEVI void set_city(int p, u64 v) const { self().sw(W_CITY_LO + p, (u32)v); self().sw(W_CITY_HI + p, (u32)(v >> 32)); }
    DEVI u64 road_lo(int p) const { return (u64)self().w(W_ROAD0 + p) | ((u64)self().w(W_ROAD1 + p) << 32); }
    DEVI u32 road_hi(int p) const { return self().w(W_ROAD2 + p); }
    DEVI int flags() const { return self().b(B_FLAGS); }
    DEVI int res(int p, int r0) const { return pb(p, P_RES + r0); }
    DEVI int total(int p) const { return res(p, 0) + res(p, 1) + res(p, 2) + res(p, 3) + res(p, 4); }
    // cold byte fields (always global): ordered card lists and the pile
    DEVI int hidden(int p, int i) const { return self().cold(B_CARDS + p * 50 + i); }
    DEVI void set_hidden(int p, int i, int v) const { self().scold(B_CARDS + p * 50 + i, v); }
    DEVI int played(int p, int i) const { return self().cold(B_CARDS + p * 50 + 25 + i); }
    DEVI void set_played(int p, int i, int v) const { self().scold(B_CARDS + p * 50 + 25 + i, v); }
    DEVI int pile(int i) const { return self().cold(B_PILE + i); }
    DEVI void set_pile(int i, int v) const { self().scold(B_PILE + i, v); }
};
// view 1: the game's record in HBM (k_masks, k_reset, k_sample_random, export/import, k_obs, tier-2 longest road)
struct St : StOps<St> {
    u32* P;           // the game's record
    long e;
    DEVI St(u32* R_, long /*N*/, long e_) : P(R_ + e_ * REC), e(e_) {}
    DEVI u32 w(int r) const { return P[r]; }
    DEVI void sw(int r, u32 v) const { P[r] = v; }
    DEVI int cold(int f) const { return ((const u8*)(P + NW))[f]; }
    DEVI void scold(int f, int v) const { ((u8*)(P + NW))[f] = (u8)v; }
    DEVI int b(int f) const { return cold(f); }
    DEVI void sb(int f, int v) const { scold(f, v); }
};
// view 2: the HOT words staged in LDS as tile[word][slot] with row stride TS (k_step); cold fields stay in HBM
template <int STRIDE>
struct StLT : StOps<StLT<STRIDE>> {
    u32* T;           // LDS tile base, already offset by the slot (lane): word r at T[r * STRIDE]
    u32* P;           // the game's record in HBM (cold fields)
    long e;
    DEVI StLT(u32* T_, u32* R_, long /*N*/, long e_) : T(T_), P(R_ + e_ * REC), e(e_) {}
    DEVI u32 w(int r) const { return T[r * STRIDE]; }
    DEVI void sw(int r, u32 v) const { T[r * STRIDE] = v; }
    DEVI int b(int f) const { return ((const u8*)(T + (NW + (f >> 2)) * STRIDE))[f & 3]; }
    DEVI void sb(int f, int v) const { ((u8*)(T + (NW + (f >> 2)) * STRIDE))[f & 3] = (u8)v; }
    DEVI int cold(int f) const { return ((const u8*)(P + NW))[f]; }
    DEVI void scold(int f, int v) const { ((u8*)(P + NW))[f] = (u8)v; }
};
typedef StLT<TS> StL;     // k_step: 64 games per tile
typedef StLT<1> StL1;     // k_reset_list: one game per wave, its hot record linear in LDS
// Transposing stage-in / stage-out of the HOT words of the 64 games whose ids sit one per lane in `e` (-1 = empty slot).
// 28 x 16 B per game, two games per pass (lanes 0..55): every load/store instruction moves 2 x 448 contiguous bytes.
DEVI void stage_in(u32* tile, const u32* __restrict__ R, int e, int lane) {
    const int half = lane >= 28 ? 1 : 0, q = lane - 28 * half;
    // all 32 loads are issued before the first LDS store: one HBM round trip instead of several (VGPRs are plentiful at
    // one wave per SIMD)
    uint4 v[32];
#pragma unroll
    for (int p = 0; p < 32; p++) {
        const int eg = __shfl(e, (2 * p + half) & 63);
        v[p] = make_uint4(0, 0, 0, 0);
        if (lane < 56 && eg >= 0) v[p] = reinterpret_cast<const uint4*>(R + (long)eg * REC)[q];
    }
#pragma unroll
    for (int p = 0; p < 32; p++) {
        const int g = 2 * p + half;
        if (lane < 56) {
            u32* t = tile + (4 * q) * TS + g;
            t[0] = v[p].x; t[TS] = v[p].y; t[2 * TS] = v[p].z; t[3 * TS] = v[p].w;
        }
    }
}
// chunks [FIRST, FIRST + NCH) of the 28 16-byte chunks of the hot record, GP = 64 / NCH games per pass: k_step writes back
// only what its wave's action type can have changed (the type is wave-uniform: sorted, padded bins) - the bitboards
// (chunks 0..6) change with settle / road / city only, the estimates (7..15) with the resource-moving types.
template <int NCH = 28, int FIRST = 0, int G = 64>
DEVI void stage_out(const u32* tile, u32* __restrict__ R, int e, int lane) {
    constexpr int TSG = G + 1;
    constexpr int GP = 64 / NCH, NP = (G + GP - 1) / GP;
    const int gi = lane / NCH, q = FIRST + lane - NCH * gi;
    const bool act = lane < GP * NCH;
    uint4 v[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const int g = p * GP + gi;
        const u32* t = tile + (4 * (act ? q : FIRST)) * TSG + (g < G ? g : 0);
        v[p].x = t[0]; v[p].y = t[TSG]; v[p].z = t[2 * TSG]; v[p].w = t[3 * TSG];
    }
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const int g = p * GP + gi;
        const int eg = __shfl(e, g & 63);
        if (act && g < G && eg >= 0) reinterpret_cast<uint4*>(R + (long)eg * REC)[q] = v[p];
    }
}

// k_step's stage-in: besides the hot records, the ACTION rows (72 B) of the wave's 64 arbitrary games.  Read
// lane-per-game, each of those 18 words is a load instruction that touches 64 different cache lines (measured: 25 us per
// wave, most of k_step); read row-wise - 7 games x 9 lanes x 8 B per instruction - every line is requested once.  The
// words go through the (still empty) tile to the owning lane's registers before the state is written into it; all global
// loads are issued up front.  (Until round 4 the previous MASK rows came in the same way, 44 of 64 B per game, for a
// "mask bit set" validation; validate mode now restates Game.validate_action from the state - action_valid - and the
// step no longer reads its own previous output.)
// EST = false: the nine estimate chunks (7..15) are left out - the wave's action type neither reads nor writes them
// (propose, end_turn, robber, knight / victory point / road building cards: 40 % of the waves) - 19 chunks, 3 games per pass.
// ROW = true (fused-sampling rollouts): the action is the one the game's previous step drew into its SIDE ROW (words 12..29,
// with the game's decision counter in word 30: five 16-byte chunks, 12 games per instruction) instead of a caller's array.
template <bool EST, int G = 64, bool ROW = false>
DEVI void stage_in_all(u32* tile, const u32* __restrict__ R, const i32* __restrict__ actions, int e, int lane,
                       int (&a)[ACTION_WORDS], const u32* __restrict__ mpk = nullptr, u32* dctr = nullptr) {
    constexpr int TSG = G + 1;
    constexpr int NCH = EST ? 28 : 19, GP = 64 / NCH, NP = (G + GP - 1) / GP;
    constexpr int NPA = ROW ? (G + 11) / 12 : (G + 6) / 7;
    const int gi = lane / NCH, q0 = lane - NCH * gi, q = (EST || q0 < 7) ? q0 : q0 + 9;
    const bool act = lane < GP * NCH;
    const int ag = ROW ? lane / 5 : lane / 9, aq = ROW ? lane - 5 * ag : lane - 9 * ag;
    uint4 v[NP];
    uint4 av[NPA];
#pragma unroll
    for (int p = 0; p < NPA; p++) {
        av[p] = make_uint4(0, 0, 0, 0);
        if constexpr (ROW) {
            const int g = p * 12 + ag, eg = __shfl(e, g & 63);
            if (lane < 60 && g < G && eg >= 0) av[p] = *reinterpret_cast<const uint4*>(mpk + (long)eg * MPK_STRIDE + ROW_ACT + 4 * aq);
        } else {
            const int g = p * 7 + ag, eg = __shfl(e, g & 63);
            if (lane < 63 && g < G && eg >= 0) {
                const uint2 t2 = *reinterpret_cast<const uint2*>(actions + (long)eg * ACTION_WORDS + 2 * aq);
                av[p].x = t2.x; av[p].y = t2.y;
            }
        }
    }
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const int g = p * GP + gi, eg = __shfl(e, g & 63);
        v[p] = make_uint4(0, 0, 0, 0);
        if (act && g < G && eg >= 0) v[p] = reinterpret_cast<const uint4*>(R + (long)eg * REC)[q];
    }
#pragma unroll
    for (int p = 0; p < NPA; p++) {
        if constexpr (ROW) {
            const int g = p * 12 + ag;
            if (lane < 60 && g < G) {
                u32* t = tile + (4 * aq) * TSG + g;
                t[0] = av[p].x; t[TSG] = av[p].y; t[2 * TSG] = av[p].z; t[3 * TSG] = av[p].w;
            }
        } else {
            const int g = p * 7 + ag;
            if (lane < 63 && g < G) { tile[(2 * aq) * TSG + g] = av[p].x; tile[(2 * aq + 1) * TSG + g] = av[p].y; }
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int sl = lane < G ? lane : G - 1;               // (lanes >= G carry no game: they read slot G-1 and are never used)
#pragma unroll
    for (int i = 0; i < ACTION_WORDS; i++) a[i] = (int)tile[i * TSG + sl];
    if constexpr (ROW) *dctr = tile[ACTION_WORDS * TSG + sl];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const int g = p * GP + gi;
        if (act && g < G) {
            u32* t = tile + (4 * q) * TSG + g;
            t[0] = v[p].x; t[TSG] = v[p].y; t[2 * TSG] = v[p].z; t[3 * TSG] = v[p].w;
        }
    }
}
// fused-sampling rollouts: the whole side row of the games with e >= 0 - new masks, the action drawn for the game's next
// step, its decision counter - through the (free again) tile: 8 lanes x 16 B per game, full 128 B lines
template <int G = 64>
DEVI void stage_out_row(u32* tile, u32* __restrict__ mpk, int e, int lane, const u32 (&m)[MASK_WORDS], const int (&an)[ACTION_WORDS], u32 dctr) {
    constexpr int TSG = G + 1;
    if (lane < G) {
#pragma unroll
        for (int i = 0; i < MASK_WORDS; i++) tile[i * TSG + lane] = m[i];
        tile[MASK_WORDS * TSG + lane] = 0;
#pragma unroll
        for (int i = 0; i < ACTION_WORDS; i++) tile[(ROW_ACT + i) * TSG + lane] = (u32)an[i];
        tile[ROW_CTR * TSG + lane] = dctr;
        tile[ROW_TAG * TSG + lane] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    const int rg = lane >> 3, rq = lane & 7;
#pragma unroll
    for (int p = 0; p < (G + 7) / 8; p++) {
        const int g = p * 8 + rg, eg = __shfl(e, g & 63);
        if (g < G && eg >= 0) {
            const u32* t = tile + (4 * rq) * TSG + g;
            uint4 val; val.x = t[0]; val.y = t[TSG]; val.z = t[2 * TSG]; val.w = t[3 * TSG];
            *reinterpret_cast<uint4*>(mpk + (long)eg * MPK_STRIDE + 4 * rq) = val;
        }
    }
}
// the reverse for the new masks: rows of the games with e >= 0, through the (free again) tile
template <int G = 64>
DEVI void stage_out_masks(u32* tile, u32* __restrict__ mpk, int e, int lane, const u32 (&m)[MASK_WORDS]) {
    constexpr int TSG = G + 1;
    constexpr int NPM = (G + 20) / 21;
    const int mg = lane / 3, mq = lane - 3 * mg;
    if (lane < G) {
#pragma unroll
        for (int i = 0; i < MASK_WORDS; i++) tile[i * TSG + lane] = m[i];
        tile[MASK_WORDS * TSG + lane] = 0;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < NPM; p++) {
        const int g = p * 21 + mg, eg = __shfl(e, g & 63);
        if (lane < 63 && g < G && eg >= 0) {
            const u32* t = tile + (4 * mq) * TSG + g;
            uint4 val; val.x = t[0]; val.y = t[TSG]; val.z = t[2 * TSG]; val.w = t[3 * TSG];
            *reinterpret_cast<uint4*>(mpk + (long)eg * MPK_STRIDE + 4 * mq) = val;
        }
    }
}

DEVI int seat_of(int seatof, int p) { return (seatof >> (2 * p)) & 3; }
DEVI int pid_at(int order, int seat) { return (order >> (2 * (seat & 3))) & 3; }
// ref: game/components/player.py:12-20
DEVI int label_of(int seatof, int me, int other) { return ((seat_of(seatof, other) - seat_of(seatof, me)) & 3) - 1; }
DEVI int player_at_label(int order, int seatof, int me, int label) { return pid_at(order, seat_of(seatof, me) + 1 + label); }
DEVI int clipi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// ------------------------------------------------------------------------------------------------ RNG
// Philox4x32-10 (Salmon et al., SC'11).  Contract: SURVEY.md 8.4 / DESIGN.md "RNG".
DEVI void philox4x32_10(u32 c0, u32 c1, u32 c2, u32 c3, u32 k0, u32 k1, u32 (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        u32 hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        u32 hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        u32 n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
struct Rng {
    u32 k0, k1, e0, e1, draws;
    u32 blk_idx, o0, o1, o2, o3;      // cached philox block (4 draws per block)
    const u32* ring;                  // contract (A): the numpy-side MT19937 stream's output ring (draw d = ring[d % MT_RING]); null: Philox
    // the four draws 4 bi .. 4 bi + 3 of the game's stream
    DEVI void block(u32 bi, u32 (&o)[4]) const {
        if (ring != nullptr) {
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = ring[(4u * bi + (u32)i) & (u32)(MT_RING - 1)];
        } else philox4x32_10(bi, 0u, e0, e1, k0, k1, o);
    }
    DEVI u32 next() {
        const u32 bi = draws >> 2;
        if (bi != blk_idx) {
            u32 o[4];
            block(bi, o);
            o0 = o[0]; o1 = o[1]; o2 = o[2]; o3 = o[3]; blk_idx = bi;
        }
        const u32 sel = draws & 3;
        draws++;
        return sel == 0 ? o0 : (sel == 1 ? o1 : (sel == 2 ? o2 : o3));
    }
    // masked rejection (numpy legacy rk_interval shape)
    DEVI u32 bounded(u32 mx) {
        if (mx == 0) return 0;
        u32 mask = 0xFFFFFFFFu >> __clz(mx);
        u32 v;
        do { v = next() & mask; } while (v > mx);
        return v;
    }
};
template <class S>
DEVI Rng rng_load(const Ctx& c, const S& s) {
    Rng r;
    u64 id = c.env_id0 + (u64)s.e;
    r.k0 = c.key0; r.k1 = c.key1; r.e0 = (u32)id; r.e1 = (u32)(id >> 32); r.draws = s.w(W_RNG); r.blk_idx = 0xFFFFFFFFu; r.o0 = r.o1 = r.o2 = r.o3 = 0;
    r.ring = c.mt != nullptr ? c.mt->np.ring : nullptr;
    return r;
}
// random.choice(seq) under contract (A): CPython's _randbelow_with_getrandbits - k = n.bit_length(), the TOP k bits of one output of the
// `random` module's generator, redrawn while >= n
DEVI int mt_choice_index(MtPair* m, int n) {
    const int k = 32 - __clz(n);
    u32 d = m->cons_py, r;
    do { r = m->py.ring[d++ & (u32)(MT_RING - 1)] >> (32 - k); } while ((int)r >= n);
    m->cons_py = d;
    return (int)r;
}
// MT19937 (Matsumoto & Nishimura 1998): one output of generator g
DEVI u32 mt_next(MtGen& g) {
    if (g.idx >= 624u) {
        for (int kk = 0; kk < 624; kk++) {
            const u32 y = (g.mt[kk] & 0x80000000u) | (g.mt[(kk + 1) % 624] & 0x7fffffffu);
            g.mt[kk] = g.mt[(kk + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g.idx = 0;
    }
    u32 y = g.mt[g.idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}

// ------------------------------------------------------------------------------------------------ estimates
struct Est { int mn[5], mx[5]; };
template <class S>
DEVI void est_load(const S& s, int o, int l, Est& E) {
    int base = W_EST + (o * 3 + l) * 3;
    u32 a = s.w(base), b = s.w(base + 1), c = s.w(base + 2);
#pragma unroll
    for (int r = 0; r < 4; r++) { E.mn[r] = (a >> (8 * r)) & 255; E.mx[r] = (b >> (8 * r)) & 255; }
    E.mn[4] = c & 255; E.mx[4] = (c >> 8) & 255;
}
template <class S>
DEVI void est_store(const S& s, int o, int l, const Est& E) {
    int base = W_EST + (o * 3 + l) * 3;
    u32 a = 0, b = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) { a |= (u32)(E.mn[r] & 255) << (8 * r); b |= (u32)(E.mx[r] & 255) << (8 * r); }
    s.sw(base, a); s.sw(base + 1, b); s.sw(base + 2, (u32)(E.mn[4] & 255) | ((u32)(E.mx[4] & 255) << 8));
}
// The visible-exchange update of one estimate entry, field = clip(field + delta, 0, total) for the selected resources
// (game.py:938-944), on the packed words directly: the ten fields are bytes of three words, and per byte
//   clip(x + d, 0, T) = min(satsub(x + max(d, 0), max(-d, 0)), T)            (x, T >= 0)
// costs a handful of word instructions as long as every byte stays below 128 (hands total <= 95; checked, with the
// field-by-field form as the fall-back).  k_step is bound by its instruction count: unpacked, an entry took ~150
// instructions and a dice roll updates twelve of them.
// Packed operands: words 0/1 (mn / mx of resources 0..3) share pw/nw/tw/mw, word 2 (mn[4] | mx[4] << 8) uses pc/nc/tc/mc;
// p = positive part of delta, n = magnitude of its negative part, t = clip bound, m = 0xFF per selected byte.
constexpr u32 H4 = 0x80808080u;
DEVI u32 swar_clip(u32 x, u32 p, u32 n, u32 t, u32 m) {
    u32 v = x + p;
    u32 d = (v | H4) - n;                                  // per byte 128 + v - n: bit 7 = (v >= n), no borrow across bytes
    v = d & ~H4 & (((d & H4) >> 7) * 255u);                // saturating v - n
    d = (v | H4) - t;
    const u32 ge = ((d & H4) >> 7) * 255u;                 // 0xFF where v >= t
    v = (t & ge) | (v & ~ge);
    return (v & m) | (x & ~m);
}
// the same for delta >= 0: min(x + p, t) per selected byte
DEVI u32 swar_addmin(u32 x, u32 p, u32 t, u32 m) {
    const u32 v = x + p;
    const u32 ge = ((((v | H4) - t) & H4) >> 7) * 255u;   // 0xFF where v >= t
    const u32 r = (t & ge) | (v & ~ge);
    return (r & m) | (x & ~m);
}
template <class S>
DEVI void est_apply(const S& s, int o, int l, u32 pw, u32 nw, u32 tw, u32 mw, u32 pc, u32 nc, u32 tc, u32 mc) {
    const int base = W_EST + (o * 3 + l) * 3;
    const u32 a = s.w(base), b = s.w(base + 1), c = s.w(base + 2);
    if ((((a + pw) | (b + pw) | (c + pc) | a | b | c | pw | pc | tw | tc | nw | nc) & H4) == 0) {
        s.sw(base, swar_clip(a, pw, nw, tw, mw)); s.sw(base + 1, swar_clip(b, pw, nw, tw, mw)); s.sw(base + 2, swar_clip(c, pc, nc, tc, mc));
        return;
    }
    Est E; est_load(s, o, l, E);                            // some byte >= 128: field by field
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const int sh = 8 * (r & 3);
        const bool sel = r < 4 ? ((mw >> sh) & 1) : (mc & 1);
        if (!sel) continue;
        const int d = r < 4 ? (int)((pw >> sh) & 255) - (int)((nw >> sh) & 255) : (int)(pc & 255) - (int)(nc & 255);
        const int t = r < 4 ? (int)((tw >> sh) & 255) : (int)(tc & 255);
        E.mx[r] = clipi(E.mx[r] + d, 0, t); E.mn[r] = clipi(E.mn[r] + d, 0, t);
    }
    est_store(s, o, l, E);
}
struct D5 { int v[5]; };
DEVI D5 d5_zero() { D5 d; d.v[0] = d.v[1] = d.v[2] = d.v[3] = d.v[4] = 0; return d; }
DEVI void d5_add(D5& d, int r0, int x) {
#pragma unroll
    for (int k = 0; k < 5; k++) d.v[k] += (k == r0) ? x : 0;
}
// ref: game/game.py:921-971.  delta per r0, `touched` = bitmask of r0 keys present in the dict, thief = -1 for none.
template <class S>
DEVI void update_estimates(const S& s, int seatof, const D5& delta, int touched, int upd, int thief) {
    int total = s.total(upd);
    if (thief < 0) {                                       // visible exchange: every other player clips (game.py:936-944)
        u32 pw = 0, nw = 0, mw = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int d = delta.v[r];
            pw |= (u32)max(d, 0) << (8 * r); nw |= (u32)max(-d, 0) << (8 * r); mw |= ((touched >> r) & 1) ? 0xFFu << (8 * r) : 0u;
        }
        const u32 p4 = (u32)max(delta.v[4], 0), n4 = (u32)max(-delta.v[4], 0), t1 = (u32)total;
        const u32 mc = ((touched >> 4) & 1) ? 0xFFFFu : 0u;
        for (int o = 0; o < 4; o++)
            if (o != upd) est_apply(s, o, label_of(seatof, o, upd), pw, nw, t1 * 0x01010101u, mw, p4 * 0x0101u, n4 * 0x0101u, t1 * 0x0101u, mc);
        return;
    }
    int total_thief = thief >= 0 ? s.total(thief) : 0;
    for (int o = 0; o < 4; o++) {
        Est E;
        if (o == upd) {
            if (thief < 0) continue;
            int sl = label_of(seatof, o, thief);
            est_load(s, o, sl, E);
#pragma unroll
            for (int r = 0; r < 5; r++) if ((touched >> r) & 1) { E.mx[r] -= delta.v[r]; E.mn[r] -= delta.v[r]; }
            est_store(s, o, sl, E);
        } else {
            int l = label_of(seatof, o, upd);
            est_load(s, o, l, E);
            if (thief < 0 || o == thief) {
#pragma unroll
                for (int r = 0; r < 5; r++) if ((touched >> r) & 1) {
                    E.mx[r] = clipi(E.mx[r] + delta.v[r], 0, total);
                    E.mn[r] = clipi(E.mn[r] + delta.v[r], 0, total);
                }
                est_store(s, o, l, E);
            } else {
                int sl = label_of(seatof, o, thief);
                Est TS;
                est_load(s, o, sl, TS);
#pragma unroll
                for (int r = 0; r < 5; r++) {
                    int cmax = E.mx[r], cmin = E.mn[r];
                    E.mx[r] = clipi(cmax, 0, total);
                    E.mn[r] = clipi(cmin - 1, 0, total);
                    if (cmax > 0) {
                        TS.mx[r] = clipi(TS.mx[r] + 1, 0, total_thief);
                        TS.mn[r] = clipi(TS.mn[r], 0, total_thief);
                    }
                }
                est_store(s, o, l, E);
                est_store(s, o, sl, TS);
            }
        }
    }
}
template <class S>
DEVI void update_estimates1(const S& s, int seatof, int r0, int d, int upd) {
    D5 dl = d5_zero();
    d5_add(dl, r0, d);
    update_estimates(s, seatof, dl, 1 << r0, upd, -1);
}
// ref: game/game.py:973-1010.  lost = per-pid0 count packed one byte each.
template <class S>
DEVI void update_estimates_monopoly(const S& s, int seatof, int mono, int r0, u32 lost) {
    int total = 0;
    for (int p = 0; p < 4; p++) if (p != mono) total += (lost >> (8 * p)) & 255;
    for (int p = 0; p < 4; p++) {
        if (p == mono) {
            for (int o = 0; o < 4; o++) if (o != p) {
                int l = label_of(seatof, o, p);
                Est E; est_load(s, o, l, E);
#pragma unroll
                for (int r = 0; r < 5; r++) if (r == r0) { E.mn[r] += total; E.mx[r] += total; }
                est_store(s, o, l, E);
            }
        } else {
            int ptotal = s.total(p), pl = (lost >> (8 * p)) & 255;
            for (int o = 0; o < 4; o++) if (o != p) {
                int l = label_of(seatof, o, p);
                Est E; est_load(s, o, l, E);
#pragma unroll
                for (int r = 0; r < 5; r++) {
                    int cmax = E.mx[r], cmin = E.mn[r];
                    if (r == r0) { cmax -= pl; cmin -= pl; }
                    E.mx[r] = clipi(cmax, 0, ptotal);
                    E.mn[r] = clipi(cmin, 0, ptotal);
                }
                est_store(s, o, l, E);
            }
        }
    }
}

// resource -> bank with the visible-resources clamp (e.g. game/game.py:197-208)
template <class S>
DEVI void pay(const S& s, int p, int r0, int n) {
    s.spb(p, P_RES + r0, s.pb(p, P_RES + r0) - n);
    s.spb(p, P_VIS + r0, max(s.pb(p, P_VIS + r0) - n, 0));
    s.sb(B_BANK + r0, s.b(B_BANK + r0) + n);
}
// ref: game/game.py:253-262
template <class S>
DEVI void update_players_go(const S& s, int order, bool left) {
    int id = s.b(B_ORDER_ID);
    id = left ? (id == 0 ? 3 : id - 1) : (id == 3 ? 0 : id + 1);
    s.sb(B_ORDER_ID, id);
    s.sb(B_GO, pid_at(order, id));
}

// ------------------------------------------------------------------------------------------------ longest road
// Longest vertex-simple path (game/game.py:843-862 + game/utils.py:3-15): directed arcs u->t over the player's
// roads exist iff u holds no opponent building; a path may END on an opponent's corner but not start on or pass it.
// The number of simple paths is heavy-tailed under random play (mean ~150 DFS expansions per call, p99.9 > 10^4,
// extremes > 10^6), so the search is two-tiered:
//   tier 1 (k_lr_finish, one wave per game): lane v enumerates the simple paths that start at corner v; busy lanes hand
//           untaken sibling subtrees to an LDS task queue whenever other lanes are idle; iteration budget (LR_BUDGET).
//   tier 2 (k_lr_heavy, 1 024-thread workgroups, 1..8 per overflowed game): the same DFS, bulk-synchronous work sharing
//           through a workgroup pool, so a dense road network gets whole CUs; the workgroup that arrives last completes the game.
// DFS state per lane: the vertex path is a 1-byte-per-level stack in LDS (children are re-derived from the
// adjacency bitmasks and the `seen` bitmask on backtrack; bit 6 = "remaining siblings were given away").
constexpr int LR_QN = 256;          // tier-1 (wave) queue entries
constexpr int LR_BUDGET = 16;       // tier-1 double-iterations per lane before the game is handed to tier 2 (lock-step; swept)
constexpr int LR_HEAVY_THREADS = 1024;
constexpr int LR_POOL = 3072;       // tier-2 workgroup pool entries
constexpr int LR_ROUND = 16;        // tier-2 iterations per bulk-synchronous round: deferred windows (throughput)
constexpr int LR_ROUND_LOCKSTEP = 4;    // ... inside a lock-step step (latency: the step waits for the deepest search; swept)

// Two kinds of search share the DFS below.
//   FULL     Game.get_longest_path: every lane v < 54 starts at its own corner (phase 1 from the start).
//   THROUGH  the longest valid path that CONTAINS a given edge (u, v) - what a newly built road can add.  While the player's
//            cached longest path is exact (LrCache), max(cached, THROUGH(new edge)) is the new longest path, and THROUGH is an
//            order of magnitude cheaper than FULL (measured with the oracle over 30 000 road builds: mean 21 vs 153 DFS
//            expansions, 0.05 % vs 0.4 % above 4 096).  One seed: phase 0 walks an arm away from u (v is marked seen from
//            the start); at EVERY arm tip the extra move SW crosses the new edge to v, phase 1 then walks the second arm.
//            SW is candidate bit 63, i.e. always the last sibling.  A path is valid unless BOTH its end corners hold an
//            opponent building (either end may be the directed path's last vertex, game.py:850-858): `ablk` remembers
//            whether the phase-0 end is blocked.
// Every lane tracks the vertex set of its best path (`bseen`): len << 54 | vertex mask is the search result.
constexpr u64 LR_SW = 1ull << 63;
constexpr u64 LR_VMASK = (1ull << 54) - 1;
constexpr u64 LRQ_PH1 = 1ull << 62, LRQ_ABLK = 1ull << 61;      // flags of a queued task, in the spare bits of its `seen`
struct Dfs { bool active; int cur, d, base, best; u64 seen, cand, bseen; int ph; bool ablk; };
struct DfsQueue { u64* seen; unsigned short* cd; int* n; int cap; };
struct LrGraph { const u64* adj; u64 blk; int sw_v; };            // adj[v] = 0 for a blocked corner; blk = blocked corners
typedef unsigned short lrstk_t;   // stack entry: vertex | sibling << 6 | mode << 12 (0 none left, 1 one sibling, 2 re-derive) | phase-0 level << 14

// A lane's best path so far: the longest, and among equally long ones the one with the LARGEST vertex mask.  The tie-break makes the
// result of a search - len << 54 | vertex mask, maximised over all lanes - independent of which lane walks which subtree: every
// longest path is some lane's candidate, so the maximum is the same whatever the work sharing did.  (With "first found" per lane
// the cached vertex set of tier 2 depended on the order in which its waves popped the pool; lengths were exact either way, but
// which later settlements invalidate the cache - and with that which requests overflow into tier 2 - differed from run to run:
// 7-11 of 65 536 games ended a deferred rollout a window or two of decisions apart, tools/deferred_reproducible.py.)
// TIE = false (tier 1, one wave: its work sharing through LDS is deterministic already): "first found" - the equal-length case
// comes up at most leaves, and the extra compare cost the DFS step 10 % (k_lr_finish 47 -> 52 us, deferred 1.16 -> 1.115 G steps/s).
template <bool TIE>
DEVI void dfs_consider(Dfs& t, int len, u64 seen) {
    if (len > t.best || (TIE && len == t.best && seen > t.bseen)) { t.best = len; t.bseen = seen; }
}
DEVI bool lrq_push(const DfsQueue& q, u64 seen, int vertex, int depth) {
    const int qi = atomicAdd(q.n, 1);
    if (qi < q.cap) { q.seen[qi] = seen; q.cd[qi] = (unsigned short)(vertex | (depth << 8)); return true; }
    atomicSub(q.n, 1);                                  // pool full: the owner keeps the sibling (an already pushed one is re-explored; harmless)
    return false;
}
// One DFS step of one lane: (at most) one backtrack followed by one descend attempt.  path: this lane's column (entry for
// level d at path[d * stride]).
template <bool TIE>
DEVI void dfs_iter(Dfs& t, const LrGraph& G, lrstk_t* path, int stride, bool donate, const DfsQueue& q) {
    if (!t.active) return;
    if (t.cand == 0) {
        if (t.d == t.base) { t.active = false; return; }
        const int child = t.cur;
        t.d--;
        const int pk = path[t.d * stride];
        const bool lvl0 = (pk >> 14) & 1;
        if (t.ph == 1 && lvl0) t.ph = 0;                // back over the new edge: v stays marked
        else t.seen &= ~(1ull << child);
        t.cur = pk & 63;
        const int mode = (pk >> 12) & 3;
        t.cand = mode == 0 ? 0ull : (mode == 1 ? (1ull << ((pk >> 6) & 63))
                                               : ((G.adj[t.cur] & ~t.seen & ~((2ull << child) - 1)) | (lvl0 ? LR_SW : 0ull)));
        if (t.cand == 0) return;
    }
    const bool sw = (t.cand & LR_VMASK) == 0;            // only the crossing move is left
    int v;
    if (sw) { v = G.sw_v; t.cand = 0; }
    else { v = __ffsll((long long)t.cand) - 1; t.cand &= t.cand - 1; }
    const int nph = sw ? 1 : t.ph;
    const bool ablk = sw ? (((G.blk >> t.cur) & 1) != 0) : t.ablk;
    const u64 nseen = t.seen | (1ull << v);
    if (nph == 1 && t.d + 1 >= t.best + (TIE ? 0 : 1) && !(ablk && ((G.blk >> v) & 1))) dfs_consider<TIE>(t, t.d + 1, nseen);
    const u64 a = (G.adj[v] & ~nseen) | (nph == 0 ? LR_SW : 0ull);
    if (a) {
        if (t.cand && donate) {                         // give the untaken siblings (<= 2 corners, and / or the crossing) away
            bool all = true;
            u64 cc = t.cand;
            while (cc) {
                if ((cc & LR_VMASK) == 0) {             // the crossing from this tip
                    cc = 0;
                    const bool ab = ((G.blk >> t.cur) & 1) != 0;
                    const u64 sseen = t.seen;            // (v is in it already)
                    if (t.d + 1 >= t.best + (TIE ? 0 : 1) && !(ab && ((G.blk >> G.sw_v) & 1))) dfs_consider<TIE>(t, t.d + 1, sseen);
                    if (G.adj[G.sw_v] & ~sseen) all &= lrq_push(q, sseen | LRQ_PH1 | (ab ? LRQ_ABLK : 0ull), G.sw_v, t.d + 1);
                } else {
                    const int sb = __ffsll((long long)cc) - 1;
                    cc &= cc - 1;
                    const u64 sseen = t.seen | (1ull << sb);
                    if (t.ph == 1) {
                        if (t.d + 1 >= t.best + (TIE ? 0 : 1) && !(t.ablk && ((G.blk >> sb) & 1))) dfs_consider<TIE>(t, t.d + 1, sseen);
                        if (G.adj[sb] & ~sseen) all &= lrq_push(q, sseen | LRQ_PH1 | (t.ablk ? LRQ_ABLK : 0ull), sb, t.d + 1);
                    } else all &= lrq_push(q, sseen, sb, t.d + 1);      // an arm tip can always cross
                }
            }
            if (all) t.cand = 0;
        }
        int entry = t.cur | (t.ph == 0 ? (1 << 14) : 0);
        if (t.cand) {
            const u64 rest = t.cand & (t.cand - 1);
            entry |= rest ? (2 << 12) : ((1 << 12) | ((__ffsll((long long)t.cand) - 1) << 6));
        }
        path[t.d * stride] = (lrstk_t)entry;
        t.cur = v; t.seen = nseen; t.d++; t.cand = a; t.ph = nph; t.ablk = ablk;
    }
}
DEVI void dfs_take(Dfs& t, const LrGraph& G, u64 seen, int cd) {
    t.seen = seen & LR_VMASK; t.ph = (seen & LRQ_PH1) ? 1 : 0; t.ablk = (seen & LRQ_ABLK) != 0;
    t.cur = cd & 255; t.d = cd >> 8; t.base = t.d;
    t.cand = (G.adj[t.cur] & ~t.seen) | (t.ph == 0 ? LR_SW : 0ull);
    t.active = true;
}
// Static split of a search over many lanes without any communication: lane `code` walks the prefix whose level-l move is
// the (code >> 2l & 3)-th candidate of that level (corners ascending, the crossing last: at most 4 moves per level), then owns
// the subtree below (base = its depth).  Every prefix node's own path length is counted by all lanes that pass it (a
// maximum: harmless); a code that names a move which does not exist leaves its lane idle (it will take shared work).
template <bool TIE>
DEVI void dfs_walk_prefix(Dfs& t, const LrGraph& G, int code, int levels) {
    for (int l = 0; l < levels && t.active; l++) {
        u64 cc = t.cand;
        for (int i = (code >> (2 * l)) & 3; i > 0 && cc; i--) cc &= cc - 1;
        if (cc == 0) { t.active = false; break; }
        const bool sw = (cc & LR_VMASK) == 0;
        const int v = sw ? G.sw_v : __ffsll((long long)cc) - 1;
        const int nph = sw ? 1 : t.ph;
        const bool ablk = sw ? (((G.blk >> t.cur) & 1) != 0) : t.ablk;
        const u64 nseen = t.seen | (1ull << v);
        if (nph == 1 && t.d + 1 >= t.best + (TIE ? 0 : 1) && !(ablk && ((G.blk >> v) & 1))) dfs_consider<TIE>(t, t.d + 1, nseen);
        t.cur = v; t.seen = nseen; t.d++; t.ph = nph; t.ablk = ablk;
        t.cand = (G.adj[v] & ~nseen) | (nph == 0 ? LR_SW : 0ull);
        if (t.cand == 0) t.active = false;
    }
    t.base = t.d;
}
DEVI void dfs_seed_full(Dfs& t, int v, u64 adjv) {
    t.active = adjv != 0; t.cur = v; t.d = 0; t.base = 0; t.best = 0; t.bseen = 0; t.ph = 1; t.ablk = false;
    t.seen = 1ull << (v & 63); t.cand = adjv;
}
DEVI void dfs_seed_through(Dfs& t, bool mine, int u, int v, u64 adju) {
    t.active = mine; t.cur = u; t.d = 0; t.base = 0; t.best = 0; t.bseen = 0; t.ph = 0; t.ablk = false;
    t.seen = (1ull << u) | (1ull << v); t.cand = (adju & ~t.seen) | LR_SW;
}
// search result: len << 54 | vertex mask of one longest path; the length of a cached result
DEVI u64 lr_pack(int best, u64 bseen) { return ((u64)best << 54) | (bseen & LR_VMASK); }
DEVI int lr_len(u64 packed_or_cache) { const u64 m = packed_or_cache & LR_VMASK; return m ? __popcll(m) - 1 : 0; }
constexpr u64 LR_OVERFLOW = ~0ull;

// The exact longest path of every player, cached in the seven spare words behind the card lists of the game's record
// (words NROWS .. REC-1; zero in a fresh record): per player the vertex mask of ONE longest valid path (54 bits; the length
// is popcount - 1) and an INVALID flag.  Exact means: what Game.get_longest_path would return now.  It stays exact while
//   - the player builds roads: max(cached, THROUGH(new edge)) (a new road only adds paths that contain it);
//   - opponents build settlements that are not on the cached path (blocking removes paths, the cached one survives);
// an opponent settlement ON the cached path, or an imported state, sets INVALID: the next need is a FULL search.
// Packing: player p < 3 in words 2p (mask bits 0..31), 2p+1 (bits 0..21: mask bits 32..53, bit 31: INVALID, bits 22..30:
// spare); player 3: word 6 (mask bits 0..31), its upper 22 mask bits and INVALID in the 27 spare bits.
constexpr int W_LRC = NROWS;
static_assert(W_LRC + 7 == REC, "the longest-path cache fills the record's tail");
constexpr u64 LRC_INVALID = 1ull << 63;
struct LrCache {
    u32 w[7];
    DEVI void load(const u32* P) {
#pragma unroll
        for (int i = 0; i < 7; i++) w[i] = P[W_LRC + i];
    }
    DEVI void store(u32* P) const {
#pragma unroll
        for (int i = 0; i < 7; i++) P[W_LRC + i] = w[i];
    }
    DEVI u64 get(int p) const {                           // mask | INVALID
        u64 r = 0;
#pragma unroll
        for (int q = 0; q < 3; q++) if (p == q) r = (u64)w[2 * q] | ((u64)(w[2 * q + 1] & 0x3FFFFFu) << 32) | ((u64)(w[2 * q + 1] >> 31) << 63);
        if (p == 3) {
            const u32 s0 = (w[1] >> 22) & 0x1FFu, s1 = (w[3] >> 22) & 0x1FFu, s2 = (w[5] >> 22) & 0x1FFu;
            r = (u64)w[6] | ((u64)(s0 | (s1 << 9) | ((s2 & 0xFu) << 18)) << 32) | ((u64)((s2 >> 4) & 1u) << 63);
        }
        return r;
    }
    DEVI void set(int p, u64 v) {
        const u32 lo = (u32)v, hi = (u32)(v >> 32) & 0x3FFFFFu, inv = (u32)(v >> 63);
#pragma unroll
        for (int q = 0; q < 3; q++) if (p == q) { w[2 * q] = lo; w[2 * q + 1] = (w[2 * q + 1] & (0x1FFu << 22)) | hi | (inv << 31); }
        if (p == 3) {
            w[6] = lo;
            w[1] = (w[1] & ~(0x1FFu << 22)) | ((hi & 0x1FFu) << 22);
            w[3] = (w[3] & ~(0x1FFu << 22)) | (((hi >> 9) & 0x1FFu) << 22);
            w[5] = (w[5] & ~(0x1FFu << 22)) | ((((hi >> 18) & 0xFu) | (inv << 4)) << 22);
        }
    }
    DEVI bool valid(int p) const { return (get(p) & LRC_INVALID) == 0; }
    // an opponent of `builder` whose cached path runs over corner cn loses it
    DEVI bool settle_invalidate(int builder, int cn) {
        bool ch = false;
#pragma unroll
        for (int o = 0; o < 4; o++) {
            const u64 v = get(o);
            if (o != builder && !(v & LRC_INVALID) && ((v >> cn) & 1)) { set(o, v | LRC_INVALID); ch = true; }
        }
        return ch;
    }
    DEVI void invalidate_all() {
#pragma unroll
        for (int o = 0; o < 4; o++) set(o, get(o) | LRC_INVALID);
    }
};
// static neighbour tables of corner v packed in two registers (loaded once per kernel): 3 x 8 bit each
DEVI void lr_load_nbr(int v, u32& nc, u32& ne) {
    nc = 0xFFFFFFu; ne = 0xFFFFFFu;
    if (v < 54) {
        nc = CORNER_NBR_C[v][0] | (CORNER_NBR_C[v][1] << 8) | (CORNER_NBR_C[v][2] << 16);
        ne = CORNER_NBR_E[v][0] | (CORNER_NBR_E[v][1] << 8) | (CORNER_NBR_E[v][2] << 16);
    }
}
// adjacency bitmask of corner v over road set (R, RH) with blocked corners BL
DEVI u64 lr_adj_of(int v, u32 nc, u32 ne, u64 R, u32 RH, u64 BL) {
    u64 m = 0;
    if (v < 54 && !((BL >> v) & 1)) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            int t = (nc >> (8 * k)) & 255, ed = (ne >> (8 * k)) & 255;
            if (t != 255) {
                bool has = ed < 64 ? ((R >> ed) & 1) : ((RH >> (ed - 64)) & 1);
                if (has) m |= 1ull << t;
            }
        }
    }
    return m;
}

struct LrWave {
    u64 adj[54];
    u64 q_seen[LR_QN];
    unsigned short q_cd[LR_QN];
    int qn;
    lrstk_t path[54][64];
};

// One search by the whole wave (all 64 lanes call it with the same arguments): the road set (R, RH) - the new edge
// included -, the blocked corners BL; through: THROUGH(u, v), else FULL.  Returns len << 54 | vertex mask, or LR_OVERFLOW
// when the iteration budget ran out (budget <= 0: unlimited).
DEVI u64 lr_wave_search(u64 R, u32 RH, u64 BL, bool through, int u, int v, LrWave& L, int budget, u32 nbr_c, u32 nbr_e,
                        unsigned long long* stat = nullptr) {
    const int lane = threadIdx.x & 63;
    const DfsQueue q{ L.q_seen, L.q_cd, &L.qn, LR_QN };
    const u64 myadj = lr_adj_of(lane, nbr_c, nbr_e, R, RH, BL);
    __builtin_amdgcn_wave_barrier();
    if (lane < 54) L.adj[lane] = myadj;
    if (lane == 0) L.qn = 0;
    __builtin_amdgcn_wave_barrier();
    const LrGraph G{ L.adj, BL, v };
    Dfs t;
    if (through) { dfs_seed_through(t, true, u, v, L.adj[u]); dfs_walk_prefix<false>(t, G, lane, 3); }     // 4^3 prefixes = the 64 lanes
    else dfs_seed_full(t, lane, myadj);
    u64 idle = __ballot(!t.active);
    int it = 0;
    bool overflow = false;
    while (true) {
        dfs_iter<false>(t, G, &L.path[0][lane], 64, idle != 0, q);
        dfs_iter<false>(t, G, &L.path[0][lane], 64, idle != 0, q);
        idle = __ballot(!t.active);
        const int qn = L.qn;
        if (idle == ~0ull && qn <= 0) break;
        if (budget > 0 && ++it > budget) { overflow = true; break; }
        if (!t.active && qn > 0) {
            const int qi = atomicSub(&L.qn, 1) - 1;
            if (qi >= 0) dfs_take(t, G, L.q_seen[qi], L.q_cd[qi]);
            else atomicAdd(&L.qn, 1);
        }
    }
    if (stat != nullptr && lane == 0) { atomicAdd(&stat[0], 1ull); atomicAdd(&stat[1], (unsigned long long)it); if (overflow) atomicAdd(&stat[2], 1ull); }
    u64 best = lr_pack(t.best, t.bseen);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const u64 o = ((u64)(u32)__shfl_xor((int)(u32)(best >> 32), off) << 32) | (u32)__shfl_xor((int)(u32)best, off);
        best = o > best ? o : best;
    }
    __builtin_amdgcn_wave_barrier();
    return overflow ? LR_OVERFLOW : best;
}
// FULL searches for the lanes with want = true (one after the other, each by the whole wave): the packed result of lane
// i's game / player in lane i (LR_OVERFLOW if the budget ran out).  Must be called by all 64 lanes of the wave.
template <class S>
DEVI u64 coop_longest_path_packed(bool want, const S& s, int pid, LrWave& L, int budget, u32 nbr_c, u32 nbr_e, unsigned long long* stat = nullptr) {
    u64 bal = __ballot(want);
    if (bal == 0) return 0;
    u64 rlo = 0, blocked = 0;
    u32 rhi = 0;
    if (want) {
        rlo = s.road_lo(pid); rhi = s.road_hi(pid);
        for (int o = 0; o < 4; o++) if (o != pid) blocked |= s.settle(o) | s.city(o);
    }
    const int lane = threadIdx.x & 63;
    u64 result = 0;
    while (bal) {
        const int src = __ffsll((long long)bal) - 1;
        bal &= bal - 1;
        const u64 R = ((u64)(u32)__shfl((int)(u32)(rlo >> 32), src) << 32) | (u32)__shfl((int)(u32)rlo, src);
        const u32 RH = (u32)__shfl((int)rhi, src);
        const u64 BL = ((u64)(u32)__shfl((int)(u32)(blocked >> 32), src) << 32) | (u32)__shfl((int)(u32)blocked, src);
        const u64 r = lr_wave_search(R, RH, BL, false, 0, 0, L, budget, nbr_c, nbr_e, stat);
        if (lane == src) result = r;
    }
    return result;
}
template <class S>
DEVI int coop_longest_path(bool want, const S& s, int pid, LrWave& L, int budget, u32 nbr_c, u32 nbr_e, unsigned long long* stat = nullptr) {
    const u64 r = coop_longest_path_packed(want, s, pid, L, budget, nbr_c, nbr_e, stat);
    return r == LR_OVERFLOW ? -1 : (int)(r >> 54);
}

// scratch of the one-wave-per-game re-deal (wave_reset_game): the game's Philox stream is generated in bulk by all 64 lanes
// (RND_WORDS consecutive draws), then one lane walks it; its shuffle arrays are LDS bytes.
constexpr int RND_WORDS = 1536;     // p99.9 of a re-deal is ~1250 draws
constexpr int TOK_HORIZON = 256;    // start offsets examined per round of the parallel number-token loop
constexpr int TOK_DRAWS = 48;       // draws one shuffle of the 18 tokens may take in that loop (mean 23; longer ones fall back)
constexpr int TOK_CHAIN = 48;       // consecutive attempts examined per round
struct ResetScratch { u32 rnd[RND_WORDS]; u8 arr[32]; u8 terr[32];
                      unsigned short endp[TOK_HORIZON]; unsigned short chain[TOK_CHAIN + 1]; u8 pm[18][64]; int tok_state[4]; };
// RNG view over the pre-generated words, falling back to inline Philox beyond them (p99.9 of a reset is ~1250 draws)
struct RngBuf {
    const u32* buf; u32 base, avail;      // buf[i] = draw number base + i
    Rng slow;
    DEVI u32 next() {
        const u32 i = slow.draws - base;
        if (i < avail) { slow.draws++; return buf[i]; }
        return slow.next();
    }
    DEVI u32 bounded(u32 mx) {
        if (mx == 0) return 0;
        const u32 mask = 0xFFFFFFFFu >> __clz(mx);
        u32 v;
        do { v = next() & mask; } while (v > mx);
        return v;
    }
};
DEVI void shuffle_bytes(u8* a, int n, RngBuf& rng) {     // np.random.shuffle on a list
    for (int i = n - 1; i >= 1; i--) {
        const int j = (int)rng.bounded((u32)i);
        const u8 t = a[i]; a[i] = a[j]; a[j] = t;
    }
}
// Board.reset + Game.reset (board.py:67-100, game.py:39-136) on LDS byte arrays and the pre-generated random words.
// The shuffles are serial by nature and run on ONE lane - except the rejection loop of the number tokens (board.py:79-81:
// re-shuffle the 18 tokens until no two red numbers touch: 8 attempts on average, geometric tail), which made the re-deal the
// longest pole of a lock-step step (a step waits for the slowest of its ~40 re-deals).  That loop runs on the whole wave:
//   (1) a shuffle started at stream offset q always reads q, q+1, ... - only HOW MANY words it takes depends on the data - so
//       every lane finds, for its share of the next TOK_HORIZON start offsets, where a shuffle started there would end;
//   (2) one lane follows these end offsets: the start offsets of the next TOK_CHAIN attempts;
//   (3) lane k replays attempt k on an identity array: the permutation that attempt applies;
//   (4) the attempts are then applied in order - one parallel gather each - until the first board without touching reds.
// Same draws, same order, same result as the serial loop (which remains the fall-back when a shuffle needs more than
// TOK_DRAWS words or the pre-generated words run out).
DEVI u32 fy_mask(int i) { return 0xFFFFFFFFu >> __clz((u32)i); }
template <class S>
DEVI void reset_part1(const S& s, RngBuf& rng, ResetScratch& sc, int hot_rows) {       // lane 0
    u8* arr = sc.arr; u8* terr = sc.terr;
    for (int r = 0; r < hot_rows; r++) if (r != W_RNG) s.sw(r, 0);
    for (int i = 0; i < 19; i++) terr[i] = (u8)(i == 0 ? 0 : (i < 4 ? 1 : (i < 8 ? 5 : (i < 12 ? 2 : (i < 15 ? 3 : 4)))));
    shuffle_bytes(terr, 19, rng);                              // board.py:72
    // board.py:25: 5 2 6 3 8 10 9 12 11 4 8 10 9 4 5 6 3 11, packed as nibbles (no constant-memory table)
    for (int i = 0; i < 16; i++) arr[i] = (u8)((0x6549A84BC9A83625ull >> (4 * i)) & 15);       // entry i in bits 4i..4i+3
    arr[16] = 3; arr[17] = 11;
    sc.tok_state[0] = (int)(rng.slow.draws - rng.base);        // stream offset of the first token shuffle
}
// all 64 lanes; returns (on every lane) whether the board is complete; sc.tok_state[0] = stream offset behind the loop
DEVI bool reset_tokens_parallel(ResetScratch& sc, int lane) {
    // per lane = tile t: its index in the spiral placement order, its neighbours, its terrain
    int pidx = 0;
#pragma unroll
    for (int i = 0; i < 19; i++) if (topo_placement(i) == lane) pidx = i;
    const u32 nbr = lane < 19 ? topo_tile_nbr_mask(lane) : 0u;
    const bool desert = lane < 19 && sc.terr[lane] == 0;
    const int dpos = __shfl(pidx, __ffsll((long long)__ballot(desert)) - 1);      // the desert takes no token
    const int tok_of_tile = pidx - (pidx > dpos ? 1 : 0);
    int off = sc.tok_state[0];
    for (int round = 0; round < 16; round++) {                                     // (bounded; the fall-back finishes)
        const int hz = min(TOK_HORIZON, RND_WORDS - TOK_DRAWS - off);            // start offsets whose TOK_DRAWS words exist
        if (hz <= 0) return false;
        // (1) end offset of a shuffle started at off + q
        for (int q = lane; q < hz; q += 64) {
            int i = 17, endq = 0xFFFF;
#pragma unroll 8
            for (int t = 0; t < TOK_DRAWS; t++) {
                const u32 w = sc.rnd[off + q + t];
                if (i >= 1 && (w & fy_mask(i)) <= (u32)i) { i--; if (i == 0) endq = q + t + 1; }
            }
            sc.endp[q] = (unsigned short)endq;
        }
        __builtin_amdgcn_wave_barrier();
        // (2) the chain of attempt starts (relative to off)
        if (lane == 0) {
            int q = 0, n = 0;
            while (n < TOK_CHAIN && q < hz) { sc.chain[n++] = (unsigned short)q; const int e = sc.endp[q]; if (e == 0xFFFF) { q = 0xFFFF; break; } q = e; }
            sc.chain[n] = (unsigned short)q;                    // where attempt n would start (0xFFFF: attempt n-1 needs the fall-back)
            sc.tok_state[1] = n;
        }
        __builtin_amdgcn_wave_barrier();
        const int n = sc.tok_state[1];
        // (3) lane k: the permutation of attempt k (an attempt without an end offset is never applied)
        if (lane < n && sc.chain[lane + 1] != 0xFFFF) {
#pragma unroll
            for (int r = 0; r < 18; r++) sc.pm[r][lane] = (u8)r;
            const int q = sc.chain[lane];
            int i = 17;
            for (int t = 0; t < TOK_DRAWS && i >= 1; t++) {
                const u32 v = sc.rnd[off + q + t] & fy_mask(i);
                if (v <= (u32)i) { const u8 x = sc.pm[i][lane]; sc.pm[i][lane] = sc.pm[v][lane]; sc.pm[v][lane] = x; i--; }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // (4) apply the attempts in order until a board has no two touching reds
        for (int k = 0; k < n; k++) {
            if (sc.chain[k + 1] == 0xFFFF) { if (lane == 0) sc.tok_state[0] = off + sc.chain[k]; __builtin_amdgcn_wave_barrier(); return false; }
            const int nv = lane < 18 ? sc.arr[sc.pm[lane][k]] : 0;
            __builtin_amdgcn_wave_barrier();
            if (lane < 18) sc.arr[lane] = (u8)nv;
            __builtin_amdgcn_wave_barrier();
            const int v = (lane < 19 && !desert) ? sc.arr[tok_of_tile] : 7;
            const bool red = lane < 19 && (v == 6 || v == 8);
            const u32 reds = (u32)__ballot(red);
            const bool clash = __ballot(red && (nbr & reds)) != 0;
            if (!clash) { if (lane == 0) sc.tok_state[0] = off + sc.chain[k + 1]; __builtin_amdgcn_wave_barrier(); return true; }
        }
        off += sc.chain[n];                                     // all examined attempts failed: the next round starts behind them
        if (lane == 0) sc.tok_state[0] = off;
        __builtin_amdgcn_wave_barrier();
    }
    return false;
}
template <class S>
DEVI void reset_part2(const S& s, RngBuf& rng, ResetScratch& sc, bool tokens_done, bool board_only = false) {   // lane 0
    u8* arr = sc.arr; u8* terr = sc.terr;
    // tile terrains in registers, indexed by constants below
    int tr[19];
#pragma unroll
    for (int t = 0; t < 19; t++) tr[t] = terr[t];
    rng.slow.draws = rng.base + (u32)sc.tok_state[0];
    bool ok = tokens_done;
    while (!ok) {                                              // board.py:79-81, 50-65 (serial form: the fall-back)
        shuffle_bytes(arr, 18, rng);
        u32 reds = 0;
        int n = 0;
#pragma unroll
        for (int i = 0; i < 19; i++) {
            const int t = topo_placement(i);
            const int v = tr[t] == 0 ? 7 : arr[n];
            if (tr[t] != 0) n++;
            if (v == 6 || v == 8) reds |= 1u << t;
        }
        ok = true;
#pragma unroll
        for (int t = 0; t < 19; t++) if (((reds >> t) & 1) && (topo_tile_nbr_mask(t) & reds)) ok = false;
    }
    {
        int n = 0;
#pragma unroll
        for (int i = 0; i < 19; i++) {                        // board.py:91-100
            const int t = topo_placement(i);
            int v;
            if (tr[t] == 0) { v = 7; s.sb(B_ROBBER, t); } else { v = arr[n]; n++; }
            s.sb(B_TILE + t, tr[t] | (v << 4));
        }
    }
    for (int i = 0; i < 9; i++) arr[i] = (u8)i;
    shuffle_bytes(arr, 9, rng);                                // board.py:84
    for (int i = 0; i < 9; i++) s.sb(B_HARB + i, arr[i]);
    // Board.reset alone (the Board() constructor, board.py:47, which runs before Game.reset deals for the first time): its draws are taken,
    // the record is not a game yet - the caller resets again (catan_reset_board_only)
    if (board_only) { s.sw(W_RNG, rng.slow.draws); return; }
    for (int i = 0; i < 4; i++) arr[i] = (u8)i;                // game.py:41
    shuffle_bytes(arr, 4, rng);                                // game.py:42
    {
        int order = 0, seatof = 0;
        for (int i = 0; i < 4; i++) { const int p = arr[i]; order |= p << (2 * i); seatof |= i << (2 * p); }
        s.sb(B_ORDER, order); s.sb(B_SEATOF, seatof);
        s.sb(B_GO, order & 3); s.sb(B_ORDER_ID, 0);
    }
    for (int r = 0; r < 5; r++) s.sb(B_BANK + r, 19);         // game.py:48-54
    for (int p = 0; p < 4; p++) { s.spb(p, P_SLEFT, 5); s.spb(p, P_CLEFT, 4); s.spb(p, P_ISECOND, 255); }
    for (int i = 0; i < 25; i++) arr[i] = (u8)(i < 14 ? C_KNIGHT : (i < 19 ? C_VP : (i < 21 ? C_YOP : (i < 23 ? C_RB : C_MONO))));
    shuffle_bytes(arr, 25, rng);                               // game.py:77
    for (int i = 0; i < 25; i++) s.set_pile(i, arr[i]);
    s.sb(B_PILE_LEN, 25);
    s.sb(B_FLAGS, F_INITIAL);
    s.sw(W_RNG, rng.slow.draws);
}
// ------------------------------------------------------------------------------------------------ masks
template <int OFF, int NBITS>
DEVI void setr(u32 (&m)[MASK_WORDS], u64 v) {      // overwrite bit range [OFF, OFF+NBITS) with the low NBITS of v
    constexpr int w0 = OFF >> 5, sh = OFF & 31;
    constexpr u64 full = NBITS >= 64 ? ~0ull : ((1ull << NBITS) - 1);
    v &= full;
    m[w0] = (m[w0] & ~(u32)(full << sh)) | (u32)(v << sh);
    if constexpr (sh + NBITS > 32) m[w0 + 1] = (m[w0 + 1] & ~(u32)(full >> (32 - sh))) | (u32)(v >> (32 - sh));
    if constexpr (sh + NBITS > 64) m[w0 + 2] = (m[w0 + 2] & ~(u32)(full >> (64 - sh))) | (u32)(v >> (64 - sh));
}
template <int OFF, int NBITS>
DEVI u64 getr(const u32 (&m)[MASK_WORDS]) {
    constexpr int w0 = OFF >> 5, sh = OFF & 31;
    constexpr u64 full = NBITS >= 64 ? ~0ull : ((1ull << NBITS) - 1);
    u64 v = (u64)m[w0] >> sh;
    if constexpr (sh + NBITS > 32) v |= (u64)m[w0 + 1] << (32 - sh);
    if constexpr (sh + NBITS > 64) v |= (u64)m[w0 + 2] << (64 - sh);
    return v & full;
}
struct Boards { u64 occ, own_bld, own_set, own_rlo, all_rlo; u32 own_rhi, all_rhi; };
template <class S>
DEVI void load_boards(const S& s, int pid, Boards& b) {
    b.occ = 0; b.own_bld = 0; b.own_set = 0; b.own_rlo = 0; b.all_rlo = 0; b.own_rhi = 0; b.all_rhi = 0;
    for (int p = 0; p < 4; p++) {
        u64 st = s.settle(p), ct = s.city(p), rl = s.road_lo(p);
        u32 rh = s.road_hi(p);
        b.occ |= st | ct; b.all_rlo |= rl; b.all_rhi |= rh;
        if (p == pid) { b.own_bld = st | ct; b.own_set = st; b.own_rlo = rl; b.own_rhi = rh; }
    }
}
// ref: game/components/corner.py:24-39 over all corners.  initial=true ignores the own-road requirement.
DEVI u64 settle_spots(const Boards& b, bool initial) {
    const u64 blocked = topo_blocked(b.occ);
    if (!initial) return ~blocked & topo_touched(b.own_rlo, b.own_rhi) & ALL54;
    return ~blocked & ALL54;
}
// ref: env/wrapper.py:368-388.  returns 5-bit card mask; yop_ok -> bank vector valid
template <class S>
DEVI int dev_card_mask(const S& s, int pid, u32& bank_bits) {
    int m = 0, banksum = 0;
    bank_bits = 0;
    for (int r = 0; r < 5; r++) { int v = s.b(B_BANK + r); banksum += v; if (v > 0) bank_bits |= 1u << r; }
    for (int cd = 0; cd < 5; cd++) {
        int k = s.pb(pid, P_HCNT + cd);
        if (k > 0 && s.b(B_BOUGHT + cd) < k && (cd != C_YOP || banksum > 0)) m |= 1 << cd;
    }
    return m;
}
// ref: env/wrapper.py:168-290
// EnvWrapper(max_proposed_trades_per_turn, max_actions_per_turn) (wrapper.py:12-18); negative = None (no limit)
struct Limits { int max_trades; int max_actions; };
template <class S>
DEVI void compute_masks(const S& s, u32 (&m)[MASK_WORDS], Limits lim) {
    // defaults: head 0 zeros, every other head all ones (wrapper.py:172-185)
    m[0] = 0xFFFFE000u;
#pragma unroll
    for (int i = 1; i < 10; i++) m[i] = 0xFFFFFFFFu;
    m[10] = 0x1Fu;
    const int flags = s.flags();
    const int pid = s.b(B_GO);
    const int ndisc = s.b(B_NDISC);
    if (ndisc > 0) {                                                       // wrapper.py:186-192
        int d = s.b(B_DISC);
        u32 rb = 0;
        for (int r = 0; r < 5; r++) if (s.res(d, r) > 0) rb |= 1u << r;
        setr<M0, 13>(m, 1u << T_DISCARD);
        setr<M11, 5>(m, rb);
        return;
    }
    // Phases that read the board - the initial placements, road building, the normal turn - share ONE call site per topology function
    // below: a wave's 64 games are in different phases, every divergent branch taken by any lane is executed by the wave, and the
    // board-to-mask functions (54-72 test / select steps each) were up to four call sites deep (topo_touched: initial road, road
    // building, settlement spots, road spots).
    const bool initial = (flags & F_INITIAL) != 0, rbuild = !initial && (flags & F_RB_ACTIVE) != 0;
    if (!initial && !rbuild) {
        if (flags & F_JUST_ROBBER) {                                           // wrapper.py:210-213, 341-351
            int seatof = s.b(B_SEATOF);
            const u64 tm = topo_tile_corners(s.b(B_ROBBER));
            u32 tg = 0;
            for (int o = 0; o < 4; o++) if (o != pid && ((s.settle(o) | s.city(o)) & tm)) tg |= 1u << label_of(seatof, pid, o);
            setr<M0, 13>(m, 1u << T_STEAL);
            setr<M6 + 3, 3>(m, tg);
            return;
        }
        if (flags & F_MUST_RESPOND) {                                          // wrapper.py:214-218, 353-365
            int tgt = s.b(B_TRADE_TGT), nr = s.b(B_TRADE_NR);
            int need[5] = { 0, 0, 0, 0, 0 };
            for (int i = 0; i < 4; i++) if (i < nr) {
                int r = s.b(B_TRADE_RECV + i) - 1;
#pragma unroll
                for (int k = 0; k < 5; k++) need[k] += (k == r) ? 1 : 0;
            }
            bool have = true;
#pragma unroll
            for (int k = 0; k < 5; k++) if (need[k] > s.res(tgt, k)) have = false;
            setr<M0, 13>(m, 1u << T_RESPOND);
            setr<M5, 2>(m, have ? 3u : 2u);
            return;
        }
        if (!(flags & F_ROLLED)) {                                             // wrapper.py:219-229
            u32 types = 1u << T_ROLL;
            if (s.pb(pid, P_NHID) > 0 && !(flags & F_PLAYED_DEV)) {
                u32 bank_bits;
                int cm = dev_card_mask(s, pid, bank_bits);
                if (cm) {
                    types |= 1u << T_PLAYDEV;
                    setr<M4, 5>(m, cm);
                    if (cm & (1 << C_YOP)) { setr<M9 + 10, 5>(m, bank_bits); setr<M10, 5>(m, bank_bits); }
                }
            }
            setr<M0, 13>(m, types);
            return;
        }
        if (lim.max_actions >= 0 && (int)s.w(W_ACTIONS) > lim.max_actions) {   // wrapper.py:232-234: only EndTurn is left
            setr<M0, 13>(m, 1u << T_ENDTURN);
            return;
        }
    }
    Boards b;
    load_boards(s, pid, b);
    int res[5] = { 0, 0, 0, 0, 0 };
    bool ini_settle = false, second = false;
    if (initial) {                                                         // wrapper.py:195-204
        const int iset = s.pb(pid, P_ISET), iroad = s.pb(pid, P_IROAD);
        ini_settle = iset == 0 || (iset == 1 && iroad == 1);
        second = iset == 2;                                                // edge.py:23-42 after_second_settlement
    } else if (!rbuild) {
#pragma unroll
        for (int r = 0; r < 5; r++) res[r] = s.res(pid, r);
    }
    const bool turn = !initial && !rbuild;
    const bool want_settle = ini_settle || (turn && res[R_WHEAT] > 0 && res[R_SHEEP] > 0 && res[R_WOOD] > 0 && res[R_BRICK] > 0);   // :238-243
    const bool want_road = (initial && !ini_settle) || rbuild || (turn && res[R_WOOD] > 0 && res[R_BRICK] > 0);                      // :206-209, 252-256
    u64 blocked = 0, touched = 0, rlo = 0;
    u32 rhi = 0;
    if (want_settle) blocked = topo_blocked(b.occ);                        // corner.py:24-39
    if ((want_settle && !ini_settle) || (want_road && !second)) touched = topo_touched(b.own_rlo, b.own_rhi);
    if (want_road) {                                                       // wrapper.py:322-339 + edge.py:23-42; bit 72 = the dummy edge
        const u64 anchors = second ? 1ull << s.pb(pid, P_ISECOND) : b.own_bld | (touched & ~b.occ);
        topo_edges_at(anchors, rlo, rhi);
        rlo &= ~b.all_rlo; rhi &= ~b.all_rhi & 0xFFu;
        if (rbuild && rlo == 0 && rhi == 0) rhi = 1u << 8;
    }
    if (initial || rbuild) {
        if (ini_settle) { setr<M0, 13>(m, 1u << T_SETTLE); setr<M1, 54>(m, ~blocked & ALL54); }
        else { setr<M0, 13>(m, 1u << T_ROAD); setr<M2, 64>(m, rlo); setr<M2 + 64, 9>(m, rhi); }
        return;
    }
    u32 types = 1u << T_ENDTURN;                                           // wrapper.py:232
    if (want_settle) {
        const u64 v = ~blocked & touched & ALL54;
        if (v && s.pb(pid, P_SLEFT) > 0) { types |= 1u << T_SETTLE; setr<M1, 54>(m, v); }
    }
    if (res[R_WHEAT] >= 2 && res[R_ORE] >= 3 && s.pb(pid, P_CLEFT) > 0 && b.own_set) {   // :245-250
        types |= 1u << T_CITY; setr<M1 + 54, 54>(m, b.own_set);
    }
    if (want_road && (rlo | rhi)) { types |= 1u << T_ROAD; setr<M2, 64>(m, rlo); setr<M2 + 64, 9>(m, rhi); }
    if (res[R_WHEAT] > 0 && res[R_SHEEP] > 0 && res[R_ORE] > 0 && s.b(B_PILE_LEN) > 0) types |= 1u << T_BUYDEV;   // :258-260
    u32 bank_bits;
    {
        int cm = dev_card_mask(s, pid, bank_bits);
        if (s.pb(pid, P_NHID) > 0 && !(flags & F_PLAYED_DEV) && cm) {                      // :262-269
            types |= 1u << T_PLAYDEV;
            setr<M4, 5>(m, cm);
            if (cm & (1 << C_YOP)) { setr<M9 + 10, 5>(m, bank_bits); setr<M10, 5>(m, bank_bits); }
        }
    }
    {                                                                                      // :271-276, 390-412
        int hb = s.pb(pid, P_HARB);
        u32 give = 0;
#pragma unroll
        for (int r = 0; r < 5; r++) {
            if ((hb & 1) && res[r] >= 3) give |= 1u << r;
            if (((hb >> (r + 1)) & 1) && res[r] >= 2) give |= 1u << r;
            if (res[r] >= 4) give |= 1u << r;
        }
        if (give && bank_bits) { types |= 1u << T_EXCHANGE; setr<M9, 5>(m, give); setr<M10, 5>(m, bank_bits); }
    }
    if (flags & F_CAN_ROBBER) {                                                            // :278-281, 308-320
        types |= 1u << T_ROBBER; setr<M3, 19>(m, topo_tiles_at(b.occ));
    }
    {                                                                                      // :283-289
        int tot = res[0] + res[1] + res[2] + res[3] + res[4];
        if (tot > 0 && (lim.max_trades < 0 || s.b(B_TRADES) < lim.max_trades)) types |= 1u << T_PROPOSE;
    }
    setr<M0, 13>(m, types);
}

__global__ __launch_bounds__(BLOCK) void k_masks(Ctx c, u32* __restrict__ mpk, Limits lim) {
    St s(c.R, c.N, (long)blockIdx.x * BLOCK + threadIdx.x);
    if (s.e >= c.N) return;
    u32 m[MASK_WORDS];
    compute_masks(s, m, lim);
#pragma unroll
    for (int i = 0; i < MASK_WORDS; i++) mpk[s.e * MPK_STRIDE + i] = m[i];
}

// packed [N][16] -> float32 [n][325] row-major (what EnvWrapper.get_action_masks returns, batched)
__global__ __launch_bounds__(BLOCK) void k_expand_masks(const u32* __restrict__ mpk, long N, long n, float* __restrict__ out, int pitch = MPK_STRIDE) {
    long i = (long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n * MASK_BITS) return;
    long e = i / MASK_BITS;
    int j = (int)(i - e * MASK_BITS);
    out[i] = (float)((mpk[e * pitch + (j >> 5)] >> (j & 31)) & 1u);
}

// ... for a list of games (games == nullptr: row j = game j): row j = the masks of game games[j].  A game that is waiting for its
// deferred step (busy != nullptr) or a negative id gets the placeholder row "only EndTurn is legal": whatever a policy draws from
// it is well-formed, and catan_step_deferred ignores it.
__global__ __launch_bounds__(BLOCK) void k_expand_masks_of(const u32* __restrict__ mpk, const i32* __restrict__ games, long rows, long n, const u8* __restrict__ busy,
                                                           float* __restrict__ out) {
    const long i = (long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= rows * MASK_BITS) return;
    const long r = i / MASK_BITS;
    const int j = (int)(i - r * MASK_BITS);
    const long e = games != nullptr ? (long)games[r] : r;
    float v;
    if (e < 0 || e >= n || (busy != nullptr && busy[e] != 0)) v = j == M0 + T_ENDTURN ? 1.0f : 0.0f;
    else v = (float)((mpk[e * MPK_STRIDE + (j >> 5)] >> (j & 31)) & 1u);
    out[i] = v;
}

// the first 11 words (325 bits) of every game's packed mask row, contiguous: what the rollout storage keeps per decision
__global__ __launch_bounds__(BLOCK) void k_copy_masks11(const u32* __restrict__ mpk, long n, u32* __restrict__ out) {
    const long i = (long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n * 11) return;
    const long e = i / 11;
    out[i] = mpk[e * MPK_STRIDE + (int)(i - e * 11)];
}
// dst[t[r]][r][:] = src[r][:] for the rows with sel[r] != 0 (one workgroup per row): the rollout collector's "append this
// game's observation / action to its own list" (RL/ppo/game_manager.py:102-133) without compacting the selected rows on the host
__global__ __launch_bounds__(256) void k_masked_row_store(unsigned char* __restrict__ dst, const unsigned char* __restrict__ src,
                                                          const long long* __restrict__ t, const unsigned char* __restrict__ sel,
                                                          long row_bytes, long step_stride_bytes) {
    const long r = blockIdx.x;
    if (!sel[r]) return;
    unsigned char* d = dst + t[r] * step_stride_bytes + r * row_bytes;
    const unsigned char* s = src + r * row_bytes;
    if ((((unsigned long long)d | (unsigned long long)s | (unsigned long long)row_bytes) & 3) == 0) {
        for (long i = threadIdx.x; i < row_bytes / 4; i += 256) reinterpret_cast<u32*>(d)[i] = reinterpret_cast<const u32*>(s)[i];
    } else if ((((unsigned long long)d | (unsigned long long)s | (unsigned long long)row_bytes) & 1) == 0) {
        for (long i = threadIdx.x; i < row_bytes / 2; i += 256) reinterpret_cast<unsigned short*>(d)[i] = reinterpret_cast<const unsigned short*>(s)[i];
    } else {
        for (long i = threadIdx.x; i < row_bytes; i += 256) d[i] = s[i];
    }
}

// ------------------------------------------------------------------------------------------------ random policy
// position of the nth (0-based) set bit of v, -1 if there is none: branch-free rank search by halves (the peel-off loop
// `v &= v - 1` ran max-over-lanes(nth) times per wave - up to 70 for an edge pick - in a kernel bound by exactly that)
DEVI int nth_set(u64 v, int nth) {
    if (nth >= __popcll(v)) return -1;
    int pos = 0;
    u32 w = (u32)v;
    int c = __popc(w);
    if (nth >= c) { nth -= c; w = (u32)(v >> 32); pos = 32; }
    c = __popc(w & 0xFFFFu); if (nth >= c) { nth -= c; w >>= 16; pos += 16; }
    c = __popc(w & 0xFFu);   if (nth >= c) { nth -= c; w >>= 8;  pos += 8; }
    c = __popc(w & 0xFu);    if (nth >= c) { nth -= c; w >>= 4;  pos += 4; }
    c = __popc(w & 0x3u);    if (nth >= c) { nth -= c; w >>= 2;  pos += 2; }
    if (nth >= (int)(w & 1u)) pos += 1;
    return pos;
}
DEVI int pick64(u64 v, u32 w) {       // uniform pick among set bits: the ((w * k) >> 32)-th
    int k = __popcll(v);
    if (k == 0) return 0;
    return nth_set(v, (int)__umulhi(w, (u32)k));
}
// DESIGN.md "random policy": philox stream 1, blocks 2*step_idx and 2*step_idx+1 -> words w0..w7
// pctr == nullptr: every game draws with the caller's step_idx (lock-step rollouts).  Otherwise game e draws with its own
// decision counter pctr[e] (advanced here) and a busy game gets the no-op action: its trajectory does not depend on when
// it is scheduled (deferred rollouts).
// A busy game whose tag equals tag_now or tag_now2 (>= 2) is released here: its step was completed on a side stream, which
// the caller has joined before this launch.
// Returns the sampled action type (-1: busy game, no action).
template <class S>
DEVI int sample_random(const Ctx& c, const S& s, const u32 (&m)[MASK_WORDS], u32 step_idx, int (&a)[ACTION_WORDS],
                       u32* __restrict__ pctr, u8* __restrict__ busy, int tag_now, int tag_now2) {
#pragma unroll
    for (int i = 0; i < ACTION_WORDS; i++) a[i] = 0;
    if (pctr != nullptr) {
        const u32 own = pctr[s.e];                          // loaded next to the busy byte: one memory round trip, not two
        int b = busy[s.e];
        if (b >= 2 && (b == tag_now || b == tag_now2)) { busy[s.e] = 0; b = 0; }
        if (b) { a[0] = -1; return -1; }
        step_idx = own;
        pctr[s.e] = step_idx + 1;
    }
    u64 id = c.env_id0 + (u64)s.e;
    u32 w[8];
    {
        u32 o[4];
        philox4x32_10(2 * step_idx, 1u, (u32)id, (u32)(id >> 32), c.key0, c.key1, o);
        w[0] = o[0]; w[1] = o[1]; w[2] = o[2]; w[3] = o[3];
    }
    int t = pick64(getr<M0, 13>(m), w[0]);
    a[0] = t;
    switch (t) {
    case T_SETTLE: a[1] = pick64(getr<M1, 54>(m), w[1]); break;
    case T_CITY: a[1] = pick64(getr<M1 + 54, 54>(m), w[1]); break;
    case T_ROAD: {
        u64 lo = getr<M2, 64>(m), hi = getr<M2 + 64, 9>(m);
        int k = __popcll(lo) + __popcll(hi);
        int nth = k ? (int)__umulhi(w[1], (u32)k) : 0;
        int nlo = __popcll(lo);
        a[2] = k == 0 ? 0 : (nth < nlo ? nth_set(lo, nth) : 64 + nth_set(hi, nth - nlo));
        break;
    }
    case T_ROBBER: a[3] = pick64(getr<M3, 19>(m), w[1]); break;
    case T_PLAYDEV:
        a[4] = pick64(getr<M4, 5>(m), w[1]);
        if (a[4] == C_MONO) a[15] = pick64(getr<M9 + 10, 5>(m), w[2]);
        else if (a[4] == C_YOP) { a[15] = pick64(getr<M9 + 15, 5>(m), w[2]); a[16] = pick64(getr<M10, 5>(m), w[3]); }
        break;
    case T_EXCHANGE: a[15] = pick64(getr<M9, 5>(m), w[1]); a[16] = pick64(getr<M10, 5>(m), w[2]); break;
    case T_PROPOSE: {
        {
            u32 o[4];
            philox4x32_10(2 * step_idx + 1, 1u, (u32)id, (u32)(id >> 32), c.key0, c.key1, o);
            w[4] = o[0]; w[5] = o[1]; w[6] = o[2]; w[7] = o[3];
        }
        int pid = s.b(B_GO);
        int hand[5], tot = 0;
#pragma unroll
        for (int r = 0; r < 5; r++) { hand[r] = s.res(pid, r); tot += hand[r]; }
        a[6] = (int)__umulhi(w[1], 3u);
        int n_give = 1 + (int)(w[2] & 1u), n_recv = 1 + (int)((w[2] >> 1) & 1u);
        if (n_give > tot) n_give = tot;
#pragma unroll
        for (int i = 0; i < 2; i++) if (i < n_give) {
            int nth = (int)__umulhi(w[3 + i], (u32)tot), r = 0;
            bool found = false;
#pragma unroll
            for (int k = 0; k < 5; k++) if (!found) { if (nth < hand[k]) { r = k; found = true; } else nth -= hand[k]; }
            a[7 + i] = r + 1;
#pragma unroll
            for (int k = 0; k < 5; k++) hand[k] -= (k == r) ? 1 : 0;
            tot--;
        }
#pragma unroll
        for (int i = 0; i < 2; i++) if (i < n_recv) a[11 + i] = 1 + (int)__umulhi(w[5 + i], 5u);
        break;
    }
    case T_RESPOND: a[5] = pick64(getr<M5, 2>(m), w[1]); break;
    case T_STEAL: a[6] = pick64(getr<M6 + 3, 3>(m), w[1]); break;
    case T_DISCARD: a[17] = pick64(getr<M11, 5>(m), w[1]); break;
    default: break;
    }
    return t;
}
// ------------------------------------------------------------------------------------------------ step
// Validate mode = EnvWrapper.step with validate_actions=True (env/wrapper.py:36-41, the wrapper's default):
// `_translate_action` (wrapper.py:114-166, :414-426, :440-486) + `Game.validate_action` (game/game.py:264-525), restated
// branch by branch FROM THE STATE - not "the mask bit is set": the reference accepts, and applies, actions its masks never
// offer (MoveRobber onto any tile whenever can_move_robber, also before the roll; ProposeTrade past the per-turn limit or
// with nothing offered; RollDice while a Road Building card is played out; the dummy edge although a real one is free;
// Year of Plenty / Monopoly whatever the bank holds; everything past max_actions_per_turn).  false = the reference raises
// and the game stays untouched.  Head values below zero are rejected (Python would wrap a negative list index: not part of
// the action space); values at or above a head's size raise in the reference.  Same rule: oracle `orc_action_is_legal`;
// pinned by tests/golden/validate_cases.npz and tools/fuzz_validate_vs_ref.py.  The action type is wave-uniform in k_step
// (sorted bins), so the switch does not diverge.
template <class S>
DEVI bool action_valid(const S& s, const int (&a)[ACTION_WORDS]) {
    const int t = a[0];
    if (t < 0 || t > 12) return false;                                     // validate_action falls off its end -> TypeError
    // ---- what raises in _translate_action, whatever the state
    int cnt[5] = { 0, 0, 0, 0, 0 };
    if (t == T_STEAL && (a[6] < 0 || a[6] > 2)) return false;              // wrapper.py:130-138
    if (t == T_PLAYDEV) {                                                  // :140-147
        if (a[4] == C_MONO && (a[15] < 0 || a[15] > 4)) return false;
        if (a[4] == C_YOP && (a[15] < 0 || a[15] > 4 || a[16] < 0 || a[16] > 4)) return false;
    }
    if (t == T_EXCHANGE && (a[15] < 0 || a[15] > 4 || a[16] < 0 || a[16] > 4)) return false;   // :148-150
    if (t == T_PROPOSE) {                                                  // :440-486 (entries behind the first 0 are never read)
        if (a[6] < 0 || a[6] > 2) return false;
        bool stop = false;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int v = a[7 + i];
            if (v == 0) stop = true;
            if (!stop) {
                if (v < 0 || v > 5) return false;
#pragma unroll
                for (int k = 0; k < 5; k++) cnt[k] += (k == v - 1) ? 1 : 0;
            }
        }
        stop = false;
#pragma unroll
        for (int i = 0; i < 4; i++) { const int v = a[11 + i]; if (v == 0) stop = true; if (!stop && (v < 0 || v > 5)) return false; }
    }
    if (t == T_RESPOND && (a[5] < 0 || a[5] > 1)) return false;            // :156-162
    if (t == T_DISCARD && (a[17] < 0 || a[17] > 4)) return false;          // :163-164
    // ---- validate_action: the discard phase first (game.py:279-303)
    const int flags = s.flags();
    const int pid = s.b(B_GO);
    if (s.b(B_NDISC) > 0) {
        if (t != T_DISCARD) return false;
        const int d = s.b(B_DISC);
        if (s.total(d) <= 7) return false;                                 // :285-286 raises (unreachable: the list only holds > 7)
        return s.res(d, a[17]) > 0;                                        // :292-300 (one card at a time: never "too many")
    }
    if (t == T_DISCARD) return false;
    const bool initial = (flags & F_INITIAL) != 0, rolled = (flags & F_ROLLED) != 0;
    const bool respond = (flags & F_MUST_RESPOND) != 0, use_dev = (flags & F_MUST_USE_DEV) != 0, just = (flags & F_JUST_ROBBER) != 0;
    // "must respond / roll first / play out the card / steal first": the builds, the purchase, exchange, proposal, EndTurn
    const bool blocked = respond || !rolled || use_dev || just;
    switch (t) {
    case T_SETTLE: {                                                       // :305-323
        if (respond || (!rolled && !initial) || use_dev || just) return false;
        if (!initial) {                                                    // can_buy_settlement :186-193
            if (s.pb(pid, P_SLEFT) <= 0) return false;
            if (!(s.res(pid, R_WHEAT) > 0 && s.res(pid, R_WOOD) > 0 && s.res(pid, R_BRICK) > 0 && s.res(pid, R_SHEEP) > 0)) return false;
        }
        if (a[1] < 0 || a[1] >= 54) return false;
        Boards b;
        load_boards(s, pid, b);
        if (!((settle_spots(b, initial) >> a[1]) & 1)) return false;       // corner.py:24-39
        if (initial) { const int k = s.pb(pid, P_ISET); return k == 0 || (k == 1 && s.pb(pid, P_IROAD) == 1); }
        return true;
    }
    case T_ROAD: {                                                         // :324-357
        const bool rb = (flags & F_RB_ACTIVE) != 0;
        if (rb && a[2] == 72) return true;                                 // :325-328: the dummy edge, nothing else is looked at
        if (!rb) {
            if (respond || (!rolled && !initial) || use_dev || just) return false;
            if (!initial && !(s.res(pid, R_WOOD) > 0 && s.res(pid, R_BRICK) > 0)) return false;   // can_buy_road :214-220
        }
        if (a[2] < 0 || a[2] >= 72) return false;                          // 72 without road building: `edges[None]` TypeError
        Boards b;
        load_boards(s, pid, b);
        u64 lo; u32 hi;                                                    // edge.py:23-42 without after_second_settlement
        topo_edges_at(b.own_bld | (topo_touched(b.own_rlo, b.own_rhi) & ~b.occ), lo, hi);
        lo &= ~b.all_rlo; hi &= ~b.all_rhi & 0xFFu;
        const int ed = a[2];
        if (!(ed < 64 ? (lo >> ed) & 1 : (hi >> (ed - 64)) & 1)) return false;
        if (rb || !initial) return true;
        const int k = s.pb(pid, P_ISET), r = s.pb(pid, P_IROAD);
        if (k == 1 && r == 0) return true;
        if (k == 2 && r == 1) {                                            // :347-352: next to the second settlement
            const int sc = s.pb(pid, P_ISECOND);
            if (sc >= 54) return false;
            u64 l2; u32 h2;
            topo_edges_at(1ull << sc, l2, h2);
            return ed < 64 ? (l2 >> ed) & 1 : (h2 >> (ed - 64)) & 1;
        }
        return false;
    }
    case T_CITY:                                                           // :358-376
        if (blocked || initial) return false;
        if (!(s.pb(pid, P_CLEFT) > 0 && s.res(pid, R_WHEAT) > 1 && s.res(pid, R_ORE) > 2)) return false;   // can_buy_city :234-238
        if (a[1] < 0 || a[1] >= 54) return false;
        return (s.settle(pid) >> a[1]) & 1;
    case T_BUYDEV:                                                         // :377-393
        if (blocked || initial) return false;
        if (!(s.res(pid, R_WHEAT) > 0 && s.res(pid, R_SHEEP) > 0 && s.res(pid, R_ORE) > 0)) return false;
        return s.b(B_PILE_LEN) > 0;
    case T_PLAYDEV: {                                                      // :394-415: no look at the bank, the dice or the resource heads
        if (respond || (flags & F_PLAYED_DEV) || initial || just) return false;
        if (a[4] < 0 || a[4] > 4) return false;
        const int k = s.pb(pid, P_HCNT + a[4]);
        return k > 0 && k != s.b(B_BOUGHT + a[4]);                         // :403-406
    }
    case T_EXCHANGE: {                                                     // :416-443, wrapper.py:428-438
        if (blocked || initial) return false;
        const int hb = s.pb(pid, P_HARB);
        const int rate = ((hb >> (a[15] + 1)) & 1) ? 2 : ((hb & 1) ? 3 : 4);
        return s.res(pid, a[15]) >= rate && s.b(B_BANK + a[16]) > 0;
    }
    case T_PROPOSE: {                                                      // :444-466: no per-turn limit, an empty offer is fine
        if (blocked || initial) return false;
#pragma unroll
        for (int k = 0; k < 5; k++) if (s.res(pid, k) < cnt[k]) return false;
        return true;
    }
    case T_RESPOND: {                                                      // :467-482
        if (!respond) return false;
        if (a[5] == 1) return true;
        const int tgt = s.b(B_TRADE_TGT), nr = s.b(B_TRADE_NR);
        int need[5] = { 0, 0, 0, 0, 0 };
        for (int i = 0; i < 4; i++) if (i < nr) {
            const int r = s.b(B_TRADE_RECV + i) - 1;
#pragma unroll
            for (int k = 0; k < 5; k++) need[k] += (k == r) ? 1 : 0;
        }
#pragma unroll
        for (int k = 0; k < 5; k++) if (need[k] > s.res(tgt, k)) return false;
        return true;
    }
    case T_ROBBER:                                                         // :483-490: ANY tile; tile >= 19: IndexError at game.py:624
        if (respond || use_dev || !(flags & F_CAN_ROBBER)) return false;
        return a[3] >= 0 && a[3] < 19;
    case T_ROLL:                                                           // :491-500 (must_use_development_card_ability is not looked at)
        return !(respond || initial || rolled || just);
    case T_ENDTURN:                                                        // :501-512
        return !(blocked || initial);
    case T_STEAL: {                                                        // :513-525
        if (respond || !just) return false;
        const int victim = player_at_label(s.b(B_ORDER), s.b(B_SEATOF), pid, a[6]);
        return ((s.settle(victim) | s.city(victim)) & topo_tile_corners(s.b(B_ROBBER))) != 0;
    }
    default: return false;
    }
}

// ref: game/game.py:138-177
// clock read that neither the scheduler nor outstanding memory operations can move work across (diagnostics only)
DEVI long long clock_fenced() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long t = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
// ref: game/game.py:138-177.  `roll` waves are a fifth of all k_step waves and were its slowest (2 300 VALU instructions,
// 8.6 us): the kernel is bound by the instruction count of its slowest wave, so this function is written for few
// instructions - one generator call site for both dice, the 19 tile bytes scanned four per word, the corner masks of the
// hit tiles from an LDS table (tct) instead of a 19-way select, the twelve estimate entries loaded together and clipped on
// the packed words.
template <class S>
DEVI int roll_dice(const S& s, Rng& rng, int order, int seatof, const u64* tct, u32* tk = nullptr) {
    long long t0 = tk ? clock_fenced() : 0;
    // np.random.randint(1, 7) twice = two masked-rejection draws on 3 bits (Rng::bounded(5)), same draw order
    // The draw loop ran as long as the wave's unluckiest lane needed draws (a quarter of them are rejected) and crossed a philox block in
    // most iterations for SOME lane: 3-4 generator evaluations per wave, and a philox block is forty quarter-rate multiplies.  Instead: the
    // two blocks that hold this game's next 5-8 draws at once (a third one when any lane of the wave found fewer than two accepted values
    // in them: 30 % of the waves), acceptance as a bit mask, the first two accepted by ffs; the game's counter advances exactly as in the
    // loop.  A lane without two accepted values among 9-12 draws (1 in 10 000) takes the loop.
    int d1 = 0, d2 = 0;
    {
        const u32 dr0 = rng.draws, b0 = dr0 >> 2, skip = dr0 & 3u;
        u32 okm = 0;
        u64 vals = 0;
        auto take = [&](int k) {
            u32 o[4];
            rng.block(b0 + (u32)k, o);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const u32 v = o[i] & 7u;
                if (4 * k + i >= (int)skip) okm |= (v <= 5u ? 1u : 0u) << (4 * k + i);
                vals |= (u64)v << (3 * (4 * k + i));
            }
        };
        take(0); take(1);
        if (__ballot(__popc(okm) < 2) != 0) take(2);
        if (__popc(okm) >= 2) {
            const int j1 = __ffs((int)okm) - 1;
            okm &= okm - 1;
            const int j2 = __ffs((int)okm) - 1;
            d1 = 1 + (int)((vals >> (3 * j1)) & 7u); d2 = 1 + (int)((vals >> (3 * j2)) & 7u);
            rng.draws = dr0 + (u32)j2 - skip + 1u;
        } else {
            for (int have = 0; have < 2;) {
                const u32 v = rng.next() & 7u;
                if (v <= 5u) { if (have == 0) d1 = 1 + (int)v; else d2 = 1 + (int)v; have++; }
            }
        }
    }
    s.sb(B_DIE1, d1); s.sb(B_DIE2, d2);
    if (tk) { const long long t1 = clock_fenced(); tk[0] = (u32)(t1 - t0); t0 = t1; }
    int roll = d1 + d2;
    if (roll == 7) {
        int n = 0;
        for (int i = 0; i < 4; i++) {
            int p = pid_at(order, i);
            if (s.total(p) > 7) { s.sb(B_DISC + n, p); n++; }
        }
        s.sb(B_NDISC, n);
        return roll;
    }
    // tiles whose number token equals the roll (at most two carry any number), robber tile excluded: the tile bytes
    // (resource | value << 4) are bytes 0..18 of five words
    static_assert(B_TILE == 0, "tile bytes start a word");
    const int robber = s.b(B_ROBBER);
    u32 tmask = 0;
#pragma unroll
    for (int w = 0; w < 5; w++) {
        const u32 x = s.w(NW + w);
        const u32 eq = ((x >> 4) & 0x0F0F0F0Fu) ^ ((u32)roll * 0x01010101u);          // byte == 0  <=>  value == roll
        u32 z = (((eq + 0x7F7F7F7Fu) & 0x80808080u) ^ 0x80808080u) >> 7;             // 1 per matching byte (bytes <= 15: no carry)
        if (w == 4) z &= 0x00010101u;                                                 // byte 19 is not a tile
        tmask |= ((z | (z >> 7) | (z >> 14) | (z >> 21)) & 15u) << (4 * w);
    }
    tmask &= ~(1u << robber);
    u32 alloc[5] = { 0, 0, 0, 0, 0 };     // per resource: one byte per pid0
    if (tmask) {
        u64 st[4], ct[4];
#pragma unroll
        for (int p = 0; p < 4; p++) { st[p] = s.settle(p); ct[p] = s.city(p); }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (tmask == 0) break;
            const int t = __ffs((int)tmask) - 1;
            tmask &= tmask - 1;
            const int r0 = (s.b(B_TILE + t) & 15) - 1;
            const u64 tm = tct[t];
            u32 add = 0;
#pragma unroll
            for (int p = 0; p < 4; p++) add |= (u32)(__popcll(st[p] & tm) + 2 * __popcll(ct[p] & tm)) << (8 * p);
#pragma unroll
            for (int k = 0; k < 5; k++) alloc[k] += (k == r0) ? add : 0u;
        }
    }
    // game.py:170-175: per resource all-or-nothing, in dict order Wood, Ore, Brick, Wheat, Sheep
    const int res_order[5] = { R_WOOD, R_ORE, R_BRICK, R_WHEAT, R_SHEEP };
    u32 okbits = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        int r0 = res_order[i];
        u32 al = alloc[r0];
        int tot = (al & 255) + ((al >> 8) & 255) + ((al >> 16) & 255) + (al >> 24);
        int bank = s.b(B_BANK + r0);
        if (tot <= bank) { okbits |= 1u << r0; if (tot) s.sb(B_BANK + r0, bank - tot); }
    }
    if (tk) { const long long t1 = clock_fenced(); tk[1] = (u32)(t1 - t0); t0 = t1; }
    // hands + estimates.  For a fixed receiving player X the five per-resource updates are sequential (the clip bound
    // is X's running hand total); different X touch disjoint estimate entries, so the player order is free.
    // Every player's estimates of X are clipped for every resource the bank could pay (game.py:172-175 calls the update
    // with zero amounts too), so all twelve entries are touched: new = min(old + add, bound) per byte.
    u32 mw = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) mw |= ((okbits >> r) & 1) ? 0xFFu << (8 * r) : 0u;
    const u32 mc = ((okbits >> 4) & 1) ? 0xFFFFu : 0u;
    u32 pw[4], tw[4], pc[4], tc[4];
#pragma unroll
    for (int X = 0; X < 4; X++) {
        int hand[5], tot = 0;
#pragma unroll
        for (int r = 0; r < 5; r++) { hand[r] = s.res(X, r); tot += hand[r]; }
        int add[5], bound[5];
#pragma unroll
        for (int r = 0; r < 5; r++) add[r] = ((okbits >> r) & 1) ? (int)((alloc[r] >> (8 * X)) & 255) : 0;
#pragma unroll
        for (int i = 0; i < 5; i++) { const int r = res_order[i]; tot += add[r]; bound[r] = tot; }   // the running hand total is the clip bound
#pragma unroll
        for (int r = 0; r < 5; r++) s.spb(X, P_RES + r, hand[r] + add[r]);       // (all twenty bytes from registers: no re-read, no branch per resource)
        pw[X] = (u32)add[0] | ((u32)add[1] << 8) | ((u32)add[2] << 16) | ((u32)add[3] << 24);
        tw[X] = (u32)bound[0] | ((u32)bound[1] << 8) | ((u32)bound[2] << 16) | ((u32)bound[3] << 24);
        pc[X] = (u32)add[4] * 0x0101u; tc[X] = (u32)bound[4] * 0x0101u;
    }
    int base[12];
    u32 ea[12], eb[12], ec[12], guard = 0;
#pragma unroll
    for (int X = 0; X < 4; X++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int o = j + (j >= X ? 1 : 0), k = X * 3 + j;
            base[k] = W_EST + (o * 3 + label_of(seatof, o, X)) * 3;
            ea[k] = s.w(base[k]); eb[k] = s.w(base[k] + 1); ec[k] = s.w(base[k] + 2);
            guard |= ea[k] | eb[k] | ec[k] | (ea[k] + pw[X]) | (eb[k] + pw[X]) | (ec[k] + pc[X]) | pw[X] | pc[X] | tw[X] | tc[X];
        }
    if ((guard & H4) == 0) {                               // every byte below 128 (always, in practice): clip on the packed words
#pragma unroll
        for (int X = 0; X < 4; X++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int k = X * 3 + j;
                s.sw(base[k], swar_addmin(ea[k], pw[X], tw[X], mw));
                s.sw(base[k] + 1, swar_addmin(eb[k], pw[X], tw[X], mw));
                s.sw(base[k] + 2, swar_addmin(ec[k], pc[X], tc[X], mc));
            }
    } else {
        for (int X = 0; X < 4; X++)
            for (int o = 0; o < 4; o++)
                if (o != X) est_apply(s, o, label_of(seatof, o, X), pw[X], 0u, tw[X], mw, pc[X], 0u, tc[X], mc);
    }
    if (tk) tk[2] = (u32)(clock_fenced() - t0);
    return roll;
}

// ref: game/game.py:817-841
template <class S>
DEVI void update_largest_army(const S& s) {
    const int order[4] = { 1, 0, 3, 2 };   // Blue, White, Red, Orange as pid0
    int max_count = 0, who = -1;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int p = order[i], k = s.pb(p, P_ARMY);
        if (k >= 3 && k > max_count) { max_count = k; who = p; }
    }
    if (who < 0) return;
    int la = s.b(B_LA_PLAYER);
    if (la == 0) { s.sb(B_LA_PLAYER, who + 1); s.sb(B_LA_COUNT, max_count); s.spb(who, P_VP, s.pb(who, P_VP) + 2); }
    else if (la == who + 1) s.sb(B_LA_COUNT, max_count);
    else if (max_count > s.b(B_LA_COUNT)) {
        s.spb(la - 1, P_VP, s.pb(la - 1, P_VP) - 2);
        s.sb(B_LA_PLAYER, who + 1); s.sb(B_LA_COUNT, max_count);
        s.spb(who, P_VP, s.pb(who, P_VP) + 2);
    }
}

struct StepCfg { int validate; int dense_reward; double win_reward; double annealing; Limits lim; int auto_reset;
                 double* reward64;              // optional unrounded rewards [n][4] (catan_set_reward_f64_buffer)
                 unsigned long long* prof;      // optional phase profile: sums / maxima over waves (atomics: coarse, perturbing)
                 u32* prof_wave;                // optional per-wave phase durations of k_step: [wave][8] ticks, plain stores
                 int prof_timeline;
                 int bin_order; };              // k_step: bins laid over the waves in BIN_ORDER_LPT order (0: in bin order)          // ... with slot 2 = the wave's START (low 32 bits of the 100 MHz wall clock) and slot 3 = where it ran
                                                //     (HW_ID | XCC_ID << 28) instead of the request-push time and the validate / switch split
// the per-wave profile buffer: PROF_WAVE_ROWS(N) rows of 8 words - k_step's waves (up to N / 16 + 17 at 16 games per wave), then at N = 65 536:
constexpr int LRF_PROF_ROW = 4128, LRF_PROF_ROWS = 2000;   // k_lr_finish's rows (one request per workgroup)
constexpr int LRH_PROF_ROW = 6128, LRH_PROF_ROWS = 1000;   // k_lr_heavy's: one row per workgroup (its first request)
constexpr long prof_wave_rows(long N) { return (N / 16 + 17 > LRH_PROF_ROW + LRH_PROF_ROWS) ? N / 16 + 17 : LRH_PROF_ROW + LRH_PROF_ROWS; }
constexpr int PROF_PHASES = 8;    // k_step: 0 stage-in, 1 validate+apply, 2 request push, 6 holder+done/reward+masks, 7 write-back;
                                  // k_reset_list: 3 philox draws per re-deal, 4 re-deals, 5 serial shuffle time
constexpr int PROF_TOTAL = 2 * PROF_PHASES + 4;   // then 14 sums, 14 counts, 14 maxima of validate+apply per action type
constexpr int PROF_WORDS = PROF_TOTAL + 42;
DEVI void prof_mark(const StepCfg& cfg, int phase, long long& t_prev) {
    if (cfg.prof_wave != nullptr) {                 // contention-free variant (k_step only): one slot per wave and phase
        const long long t = wall_clock64();
        if ((threadIdx.x & 63) == 0 && !(cfg.prof_timeline && phase == 2))
            cfg.prof_wave[((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + phase] = (u32)(t - t_prev);
        t_prev = wall_clock64();
        return;
    }
    if (cfg.prof == nullptr) return;
    long long t = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        unsigned long long dt = (unsigned long long)(t - t_prev);
        atomicAdd(&cfg.prof[phase], dt);
        atomicMax(&cfg.prof[PROF_PHASES + phase], dt);
    }
    t_prev = wall_clock64();
}

// ctr: [4], [5] lengths of the two tier-1 request lists (k_lr_finish); [6] the may-end list of a lock-step step (pend.spec);
// [8 + 4 sa]: tier-2 requests of slot sa (k_lr_heavy), [9 + 4 sa] .. [11 + 4 sa]: its re-deal lists 0..2 (k_reset_list /
// k_install_list); [16 + 18 b ..]: the two sets of games-per-bin counts of the sort.
// busy[e] != 0: game e is waiting for the slow path (longest-road completion or re-deal); it takes no action until the
// slow path has run (same step in lock-step mode, end of the window in deferred mode).
// Slow-path hand-off (all device arrays).
//   tier-1 longest-road requests go to request list `fa` (two lists, so that the list of one iteration can be worked off on
//   a side stream while the next iteration fills the other); k_step marks those games busy with `ftag`;
//   tier-2 requests and finished games go to the lists of window slot `sa`: heavy[sa], resets[sa][0] (games that ended in
//   k_step; in a deferred window also those that ended in k_lr_finish), resets[sa][1] (games that ended in the tier-2
//   completion) and resets[sa][2] (lock-step steps: games that ended in k_lr_finish - the re-deals of list 0 start right
//   behind k_step); those games are marked busy with `stag`.
//   Tags: 1 = the kernel that completes the game clears it (lock-step: everything on one stream); >= 2 = the sampler
//   clears it at a point fixed by the schedule (deferred rollouts: tier 1 two iterations later, slot `sa` two windows
//   later), never by when a side stream happens to finish.
struct Pending { u32* ctr; u64* req[3]; u64* heavy[2]; u64* heavy2; u8* type; u8* who; u64* len; u32* arrive; i32* resets[2][3]; u8* busy;
                 u64* spec;                 // lock-step steps: the longest-road requests of games that this step may end (ctr[6])
                 i32* lists;                // the sort: game ids per action-type bin, three sets of [NBINS][N] (set s, bin b, rank r at (s * NBINS + b) * N + r)
                 int bsel;                  // which bin-count set (ctr[16 + NBINS * bsel ..]) and list set this pass reads
                 int bclear;                // the set whose counts k_step zeroes for a later pass (lock-step: the other of two; fused-sampling rollouts: pass + 2 of three)
                 int fa, ftag, sa, stag;
                 // Fused-sampling deferred rollouts (sample != 0): whoever completes a game's step - k_step for most, k_lr_finish, the
                 // tier-2 completion or the re-deal for the rest - draws the game's NEXT action from the new masks (random policy, the
                 // game's own decision counter) into the game's side row and enqueues the game for the pass it plays again in: no
                 // sampling / sorting kernel is left on the critical path, and a game that is not in a pass's lists simply does not
                 // play in it (no busy tags to clear).
                 int sample;
                 int lrq_clear;             // k_step: the tier-1 request list to empty for the passes that follow (-1: none)
                 int bnext;                 // k_step: the set the games it completes are appended to (the next pass)
                 int brel;                  // slow-path completions: >= 0 the set of the pass their games return in (tier 1: pass + 2);
                                            // < 0: the window's release list resets[sa][2] (tier 2 / re-deals: k_release_window enqueues them)
                 // Fused-sampling rollouts keep every bin's list as `nsub` SUB-LISTS with a counter each (bctr, one 128-byte line per counter):
                 // k_step's waves reserve their ranges on sub-list (wave index % nsub), the one-lane producers on (game id % nsub), and the
                 // next pass reads a bin as the concatenation of its sub-lists.  With ONE counter per bin the ~2 000 waves of a launch queue up
                 // on 17 addresses (~9 us per 1 000 same-address atomics: tools/native/atomic_contention_probe.hip) - round 4's fused loop.
                 // nsub = 1 outside those rollouts (the lists' layout is then the plain [set][bin][N]).
                 u32* bctr;                 // [BIN_SETS][NBINS][nsub][BCTR_PAD]
                 int nsub;
               };
constexpr int CTR_WORDS = 96;
constexpr int BIN_SETS = 4;                  // bin-count / list sets (fused-sampling rollouts rotate period + 2 of them)
DEVI int lrq_ctr(int fl) { return fl < 2 ? 4 + fl : 7; }     // length of tier-1 request list fl (three lists in fused-sampling rollouts)
// Sort bins: 0..12 = the action types, 13..16 = play_dev with card 1..4 (card 0 stays in bin T_PLAYDEV: the five cards run
// five different code paths, and the launch lasts as long as its slowest wave), NBINS-1 = no-op / busy / padding.
constexpr int NBINS = 18, BIN_NOOP = NBINS - 1;
static_assert(16 + BIN_SETS * NBINS <= CTR_WORDS, "the sets of bin counts live in ctr[16 ..]");
constexpr int BCTR_PAD = 32;                 // words between two sub-list counters (a 128-byte line each)
constexpr int MAX_SUBS = 16;
DEVI u32* sub_ctr(const Pending& pend, int set, int bin, int sub) { return pend.bctr + (((long)set * NBINS + bin) * pend.nsub + sub) * BCTR_PAD; }
DEVI i32* sub_list(const Pending& pend, int set, int bin, int sub, long N) { return pend.lists + (((long)set * NBINS + bin) * pend.nsub + sub) * N; }
DEVI int bin_of(int t, int card) {
    if (t < 0 || t > 12) return BIN_NOOP;
    if (t != T_PLAYDEV) return t;
    card = min(max(card, 0), 4);                           // the clamp k_step applies to an unvalidated card index
    return card >= 1 ? 12 + card : t;
}
// The order in which the bins take the launch's waves (StepCfg::bin_order): workgroups start roughly in index order over a ramp of
// several microseconds, so the bins whose waves last longest go first (longest processing time first: settle, roll, respond, steal ...)
// and the short ones - propose, end_turn - start last; wave durations per bin: profiles/r05_k_step_timeline.txt.  The no-op bin (its waves
// return at once) stays last.
__device__ constexpr int BIN_ORDER_LPT[18] = { 0, 9, 7, 11, 16, 2, 14, 3, 5, 4, 13, 1, 12, 8, 15, 6, 10, 17 };
DEVI int type_of_bin(int bin) { return bin <= 12 ? bin : (bin < BIN_NOOP ? T_PLAYDEV : -1); }

// Fused-sampling rollouts, ONE lane (the slow-path completions: k_lr_finish, the tier-2 completion, the re-deal): the game's
// next action from its new masks `m` - random policy, decision index = the counter in the game's side row (k_step advanced it
// when it applied the action whose step completes here) - written into the side row, and the game enqueued: into bin set
// pend.brel (the pass it returns in), or into the window's release list (k_release_window enqueues those).
template <class S>
DEVI void sample_enqueue_lane(const Ctx& c, const S& s, const u32 (&m)[MASK_WORDS], const Pending& pend, u32* __restrict__ mpk) {
    const long e = s.e;
    u32* row = mpk + e * MPK_STRIDE;
    int a[ACTION_WORDS];
    const int t = sample_random(c, s, m, row[ROW_CTR], a, (u32*)nullptr, (u8*)nullptr, 0, 0);
    uint4* dst = reinterpret_cast<uint4*>(row + ROW_ACT);
#pragma unroll
    for (int i = 0; i < 4; i++) dst[i] = make_uint4((u32)a[4 * i], (u32)a[4 * i + 1], (u32)a[4 * i + 2], (u32)a[4 * i + 3]);
    *reinterpret_cast<uint2*>(row + ROW_ACT + 16) = make_uint2((u32)a[16], (u32)a[17]);
    if (pend.brel >= 0) {
        const int bin = bin_of(t, a[4]), sub = (int)(e & (pend.nsub - 1));
        const u32 rank = atomicAdd(sub_ctr(pend, pend.brel, bin, sub), 1u);
        sub_list(pend, pend.brel, bin, sub, c.N)[rank] = (i32)e;
    } else {
        const u32 rank = atomicAdd(&pend.ctr[11 + 4 * pend.sa], 1u);
        pend.resets[pend.sa][2][rank] = (i32)e;
    }
}

struct StepCfg;
DEVI void prof_mark(const StepCfg& cfg, int phase, long long& t_prev);
struct StepScratch { LrWave lr; };

// Everything of a step that needs the longest-road length: the holder logic of game/game.py:864-919, done/rewards
// (env/wrapper.py:85-112), auto-reset (RL/ppo/game_manager.py:112-113) and the next legal-action masks.
// Must be called by all 64 lanes; `doit` selects the lanes it applies to.
// LR = 0: no lane has a longest-road update; LR = 1: lanes may have one whose length `len` came from the cache (k_step,
// lane per game; "the holder's road was cut" cannot happen then: an exact cached length equals the holder's count);
// LR = 2: the slow-path kernels (one game per wave, `lc` = the game's cache in the doit lane): the rare cut case takes the
// other players' lengths from the cache, or searches (whole wave) and caches them.
template <int LR, class S>
DEVI void finish_step(const Ctx& c, const S& s, StepScratch* scratch, const StepCfg& cfg, int lane, bool doit, int type,
                      int lr_who, int len, float* __restrict__ reward, u8* __restrict__ done, u32* __restrict__ mpk,
                      long long& tprof, u32 nbr_c, u32 nbr_e, const Pending& pend, int rlist, bool clear_busy,
                      u32* m_out = nullptr, bool* m_valid = nullptr, LrCache* lc = nullptr) {
    const long e = s.e;
    if constexpr (LR != 0) {
        bool cut = false;
        int holder = 0, hcount = 0;
        if constexpr (LR == 2) prof_mark(cfg, 2, tprof);
        if (doit && lr_who >= 0) {
            s.spb(lr_who, P_CURLP, len);
            holder = s.b(B_LR_PLAYER); hcount = s.b(B_LR_COUNT);
            if (holder == 0) {
                if (len >= 5) { s.sb(B_LR_PLAYER, lr_who + 1); s.sb(B_LR_COUNT, len); s.spb(lr_who, P_VP, s.pb(lr_who, P_VP) + 2); }
            } else if (holder == lr_who + 1) {
                if (hcount > len) cut = true; else s.sb(B_LR_COUNT, len);
            } else if (len > hcount) {
                s.spb(holder - 1, P_VP, s.pb(holder - 1, P_VP) - 2);
                s.spb(lr_who, P_VP, s.pb(lr_who, P_VP) + 2);
                s.sb(B_LR_PLAYER, lr_who + 1); s.sb(B_LR_COUNT, len);
            }
        }
        if constexpr (LR == 2) prof_mark(cfg, 5, tprof);
        if constexpr (LR == 2) {
        if (__ballot(cut)) {                                               // game.py:880-912 (rare)
            int max_len = len, player = lr_who;
            bool tied = false;
            for (int o = 0; o < 4; o++) {                                  // White, Blue, Orange, Red (game.py:886)
                const bool mine = cut && o != lr_who;
                const bool need = mine && !lc->valid(o);
                const u64 found = coop_longest_path_packed(need, s, o, scratch->lr, 0, nbr_c, nbr_e);
                if (need) lc->set(o, found & LR_VMASK);
                if (mine) {
                    const int pl = lr_len(lc->get(o));
                    if (pl == max_len) tied = true;
                    else if (pl > max_len) { max_len = pl; tied = false; player = o; }
                }
            }
            if (cut) {
                if (max_len >= 5) {
                    if (tied) {
                        if (player == lr_who) s.sb(B_LR_COUNT, len);
                        else { s.sb(B_LR_PLAYER, 0); s.sb(B_LR_COUNT, 0); s.spb(lr_who, P_VP, s.pb(lr_who, P_VP) - 2); }
                    } else {
                        s.sb(B_LR_PLAYER, player + 1); s.sb(B_LR_COUNT, max_len);
                        s.spb(player, P_VP, s.pb(player, P_VP) + 2);
                        s.spb(lr_who, P_VP, s.pb(lr_who, P_VP) - 2);
                    }
                } else { s.sb(B_LR_PLAYER, 0); s.sb(B_LR_COUNT, 0); s.spb(lr_who, P_VP, s.pb(lr_who, P_VP) - 2); }
            }
        }
        }
    }
    if constexpr (LR == 2) prof_mark(cfg, 3, tprof);                       // (k_lr_finish profile: holder logic)
    // ---- done / rewards (wrapper.py:85-112)
    bool want_reset = false;
    if (doit) {
        const int dict_order[4] = { 1, 3, 2, 0 };                          // Blue, Red, Orange, White (game.py:18-23)
        int vps[4];
#pragma unroll
        for (int p = 0; p < 4; p++) vps[p] = s.pb(p, P_VP);
        int winner = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) if (vps[dict_order[i]] >= 10) winner = dict_order[i] + 1;
        bool dn = winner != 0 && type >= 0;
        // the reference shapes rewards in Python floats (doubles) and rounds to fp32 once, when the rollout tensors are built
        // (process_batch.py:63): same operations in the same order in double here, one rounding at the store
        double rw[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            double r = 0.0;
            if (type >= 0) {
                if (cfg.dense_reward) {
                    r += (double)(5 * (vps[p] - s.b(B_CURVP + p)));
                    if (type == T_PLAYDEV) r += 5.0;
                    if (type == T_ROBBER) r += 1.0;
                    if (type == T_DISCARD) r -= 0.3;
                    if (type == T_CITY) r += 2.5;
                    r *= cfg.annealing;
                }
                s.sb(B_CURVP + p, vps[p]);
                if (dn && winner == p + 1) r += cfg.win_reward;
            }
            rw[p] = r;
        }
        *reinterpret_cast<float4*>(reward + s.e * 4) = make_float4((float)rw[0], (float)rw[1], (float)rw[2], (float)rw[3]);   // one 16 B store per game
        if (cfg.reward64) {
            *reinterpret_cast<double2*>(cfg.reward64 + s.e * 4) = make_double2(rw[0], rw[1]);
            *reinterpret_cast<double2*>(cfg.reward64 + s.e * 4 + 2) = make_double2(rw[2], rw[3]);
        }
        if (dn) s.sb(B_WINNER, winner);
        done[s.e] = dn ? 1 : 0;
        want_reset = dn && cfg.auto_reset;
        // RL/ppo/game_manager.py:112-113: the finished game is reset by k_reset_list (one wave per game: winning moves
        // cluster in a few action-type bins, inline resets would serialise inside those waves), which also writes its masks
        if (want_reset) pend.resets[pend.sa][rlist][atomicAdd(&pend.ctr[9 + 4 * pend.sa + rlist], 1u)] = (i32)s.e;
        if (want_reset) pend.busy[s.e] = (u8)pend.stag;
        else if (lr_who >= 0 && clear_busy) pend.busy[s.e] = 0;
    }
    if constexpr (LR == 2) prof_mark(cfg, 4, tprof);                       // (k_lr_finish profile: done / rewards)
    // ---- next legal-action masks (env/wrapper.py:168-290), from the LDS tile
    if (doit && !want_reset) {
        u32 m[MASK_WORDS];
        compute_masks(s, m, cfg.lim);
        if (m_out != nullptr) {                       // the caller stores the rows (k_step: row-wise through LDS)
#pragma unroll
            for (int i = 0; i < MASK_WORDS; i++) m_out[i] = m[i];
            *m_valid = true;
        } else {
#pragma unroll
            for (int i = 0; i < MASK_WORDS; i++) mpk[e * MPK_STRIDE + i] = m[i];
        }
    }
    prof_mark(cfg, 6, tprof);
}

// The fused env step.  One wave = 64 games; the wave's HOT state rows are staged in LDS as tile[row][lane]
// (112 coalesced 256 B row loads, conflict-free LDS columns), the whole step runs out of LDS - translate + validate,
// apply_action, longest road (wave-cooperative), done/rewards, auto-reset of finished games, next legal-action
// masks - and the tile is written back once.  With 65 536 games there is exactly one wave per SIMD, so the launch
// time is the dependent-latency chain of a single wave: LDS (~64 cycles) instead of HBM/L2 (~200-900 cycles) per hop.
// actions: int32 [n][18]; reward: float [n][4] (index PlayerId-1); done: u8 [n]; err: [1] invalid-action
// counter; mpk: packed masks [N][16], rewritten with the masks of the new state (validation restates Game.validate_action from the state).
// G = games per wave (64, 32 or 16; lanes >= G carry no game and only help with the row-wise transfers): with G < 64 there
// are 64 / G waves per SIMD at 65 536 games, each with a tile of ROWS_HOT x (G + 1) words, so that one wave's transfers
// overlap the others' dependent-instruction chains (with one wave per SIMD the HBM is idle while the step computes).
// SAMPLE (fused-sampling deferred rollouts): the action comes from the game's side row, the step advances the game's decision
// counter, and for every game it completes it draws the NEXT action from the new masks (still in registers) and appends the
// game to the next pass's bin lists - the sampler / sort kernel and its re-read of the masks are gone from the pass.
// WPB waves per workgroup, every wave with a tile of its own and no workgroup-level synchronisation.  WPB = 4 (the default for 64 games
// per wave): the hardware spreads a workgroup's waves over the four SIMDs of its CU and the 116 KB of LDS admit one workgroup per CU,
// so every working wave has a SIMD to itself.  As 1 041 one-wave workgroups the dispatcher used 800 of the 1 024 SIMDs and put two to
// four waves on 208 of them (tools/step_timeline.py, profiles/r05_k_step_timeline.txt): the launch lasted as long as those.
template <int G, bool SAMPLE = false, int WPB = 1>
__global__ __launch_bounds__(64 * WPB) void k_step(Ctx c, const i32* __restrict__ actions, u32* __restrict__ mpk,
                                             float* __restrict__ reward, u8* __restrict__ done,
                                             u32* __restrict__ err, StepCfg cfg, Pending pend, const u32* __restrict__ bins) {
    constexpr int TSG = G + 1;
    typedef StLT<TSG> StG;
    __shared__ u32 tile_all[WPB][ROWS_HOT * TSG];
    __shared__ u64 tct_all[WPB][20];                        // corner mask per tile: a per-lane tile index costs one LDS read
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int wv = (int)blockIdx.x * WPB + wib;             // this wave's position among the sorted waves
    u32* const tile = tile_all[wib];
    u64* const tct = tct_all[wib];
    if constexpr (SAMPLE) {
        if (wv == 0)                                                                 // the sub-list counters of a later pass (nobody appends to that set yet)
            for (int i = lane; i < NBINS * pend.nsub; i += 64) pend.bctr[((long)pend.bclear * NBINS * pend.nsub + i) * BCTR_PAD] = 0;
        if (wv == 0 && lane == 0 && pend.lrq_clear >= 0) pend.ctr[lrq_ctr(pend.lrq_clear)] = 0;   // ... and the next group's tier-1 request list (its last reader is done)
    } else if (wv == 0 && lane < NBINS) pend.ctr[16 + NBINS * pend.bclear + lane] = 0;     // the bin counts of a later pass (nobody appends to that set yet)
    // Wave w takes the sorted positions 64w .. 64w+63.  The sort is never materialised: the sampler / k_classify left the
    // game ids in one list per bin, every bin occupies ceil(count / 64) waves (type-pure waves), and a wave finds its bin
    // and offset from the 18 counts.
    int bin = -1, cnt = 0, first = 0;
    u32 subc[SAMPLE ? MAX_SUBS : 1];                        // SAMPLE: the lengths of the wave's bin's sub-lists (wave-uniform)
    if constexpr (SAMPLE) {
        // lane b sums bin b's sub-list counters (all loads in flight together); the 18 totals and then the chosen bin's lengths are broadcast
        u32 mine[MAX_SUBS], tot = 0;
#pragma unroll
        for (int j = 0; j < MAX_SUBS; j++) mine[j] = (lane < NBINS && j < pend.nsub) ? *sub_ctr(pend, pend.bsel, lane, j) : 0u;
#pragma unroll
        for (int j = 0; j < MAX_SUBS; j++) tot += mine[j];
        const int pos = wv * G;
        int start = 0;
#pragma unroll
        for (int kk = 0; kk < NBINS; kk++) {
            const int k = cfg.bin_order ? BIN_ORDER_LPT[kk] : kk;
            const int ck = __builtin_amdgcn_readlane((int)tot, k), len = (ck + G - 1) & ~(G - 1);
            if (bin < 0 && pos < start + len) { bin = k; cnt = ck; first = pos - start; }
            start += len;
        }
        if (bin < 0) return;                               // behind the last bin
#pragma unroll
        for (int j = 0; j < MAX_SUBS; j++) subc[j] = (u32)__builtin_amdgcn_readlane((int)mine[j], bin);
    } else {
        const int pos = wv * G;
        int start = 0;
#pragma unroll
        for (int kk = 0; kk < NBINS; kk++) {
            const int k = cfg.bin_order ? BIN_ORDER_LPT[kk] : kk;
            const int ck = (int)bins[k], len = (ck + G - 1) & ~(G - 1);
            if (bin < 0 && pos < start + len) { bin = k; cnt = ck; first = pos - start; }
            start += len;
        }
        if (bin < 0) return;                               // behind the last bin
    }
    const int rnk = first + lane;
    long e = 0x7fffffffL;
    if (lane < G && rnk < cnt) {
        if constexpr (SAMPLE) {                             // position rnk of the bin = position `off` of sub-list `sub`
            int sub = 0;
            u32 off = (u32)rnk;
#pragma unroll
            for (int j = 0; j < MAX_SUBS - 1; j++) if (sub == j && j + 1 < pend.nsub && off >= subc[j]) { off -= subc[j]; sub = j + 1; }
            e = (long)sub_list(pend, pend.bsel, bin, sub, c.N)[off];
        } else e = (long)pend.lists[((long)pend.bsel * NBINS + bin) * c.N + rnk];
    }
    long long tprof = (cfg.prof || cfg.prof_wave) ? wall_clock64() : 0;
    if (cfg.prof_wave != nullptr && cfg.prof_timeline && lane == 0) cfg.prof_wave[(long)wv * 8 + 2] = (u32)tprof;
    const bool live = e < c.n;
    // the last bin = explicit no-op (negative type: frozen game) or a busy game (the sampler gives those the no-op): none
    // of them touches its record
    int type = live ? type_of_bin(bin) : -1;
    if (live && type < 0) {
        *reinterpret_cast<float4*>(reward + e * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cfg.reward64) { double2* r64 = reinterpret_cast<double2*>(cfg.reward64 + e * 4); r64[0] = make_double2(0.0, 0.0); r64[1] = make_double2(0.0, 0.0); }
        done[e] = 0;
    }
    if (__ballot(type >= 0) == 0) return;
    if (__ballot(type == T_ROLL) != 0 && lane < 19) tct[lane] = topo_tile_corners(lane);
    u32 nbr_c, nbr_e;
    lr_load_nbr(lane, nbr_c, nbr_e);
    int a[ACTION_WORDS];
    // what the wave's action type can touch (wave-uniform: sorted, padded bins): bitboards are written by settle / road /
    // city only; the estimates are read and written by the resource-moving types only
    const bool t_board = type == T_SETTLE || type == T_ROAD || type == T_CITY;
    const bool t_est = t_board || type == T_BUYDEV || bin == 12 + C_YOP || bin == 12 + C_MONO || type == T_EXCHANGE || type == T_RESPOND ||
                       type == T_ROLL || type == T_STEAL || type == T_DISCARD;
    const bool w_board = __ballot(type >= 0 && t_board) != 0, w_est = __ballot(type >= 0 && t_est) != 0;
    u32 dctr = 0;                                           // SAMPLE: the game's decision counter (side row)
    if (w_est) stage_in_all<true, G, SAMPLE>(tile, c.R, actions, type >= 0 ? (int)e : -1, lane, a, mpk, &dctr);
    else stage_in_all<false, G, SAMPLE>(tile, c.R, actions, type >= 0 ? (int)e : -1, lane, a, mpk, &dctr);
    __builtin_amdgcn_wave_barrier();
    StG s(tile + (lane < G ? lane : G - 1), c.R, c.N, e);
    prof_mark(cfg, 0, tprof);
    bool rejected = false;
    if (cfg.validate && type >= 0 && !action_valid(s, a)) { atomicAdd(err, 1u); type = -1; rejected = true; }
    if (rejected) {                                   // an illegal action leaves the game untouched (reward 0, not done)
        *reinterpret_cast<float4*>(reward + e * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cfg.reward64) { double2* r64 = reinterpret_cast<double2*>(cfg.reward64 + e * 4); r64[0] = make_double2(0.0, 0.0); r64[1] = make_double2(0.0, 0.0); }
        done[e] = 0;
    }
    // clamp indices so that an unvalidated bad action cannot touch memory outside the game's rows
    a[1] = min(max(a[1], 0), 53); a[2] = min(max(a[2], 0), 72); a[3] = min(max(a[3], 0), 18);
    a[4] = min(max(a[4], 0), 4); a[6] = min(max(a[6], 0), 2);
    a[15] = min(max(a[15], 0), 4); a[16] = min(max(a[16], 0), 4); a[17] = min(max(a[17], 0), 4);

#ifdef CATAN_FINE_PROF
    prof_mark(cfg, 4, tprof);
#endif
    const int order = s.b(B_ORDER), seatof = s.b(B_SEATOF);
    const int pid = s.b(B_GO);
    int flags = s.flags();
    int lr_who = -1, lr_edge = 127, lr_len_now = 0;         // longest-road update: for whom, the new edge (127: none), cached length
    bool lr_inline = false;                                 // ... whose length the cache already knows
    const long long t_sw0 = cfg.prof_wave ? clock_fenced() : 0;

    switch (type) {
    case T_SETTLE: {                                                       // game.py:530-555, 195-212
        int cn = a[1];
        if (!(flags & F_INITIAL)) { pay(s, pid, R_WHEAT, 1); pay(s, pid, R_SHEEP, 1); pay(s, pid, R_WOOD, 1); pay(s, pid, R_BRICK, 1); }
        s.set_settle(pid, s.settle(pid) | (1ull << cn));                   // board.py:178-184
        int slot = CORNER_HSLOT[cn];
        if (slot != 255) {
            int hid = s.b(B_HARB + slot);
            // harbour id -> resource (board.py:29-33): 0 Ore, 1 Sheep, 2 Wheat, 3 Wood, 4 Brick, 5..8 generic
            int bitpos = hid == 0 ? R_ORE + 1 : hid == 1 ? R_SHEEP + 1 : hid == 2 ? R_WHEAT + 1 : hid == 3 ? R_WOOD + 1 : hid == 4 ? R_BRICK + 1 : 0;
            s.spb(pid, P_HARB, s.pb(pid, P_HARB) | (1 << bitpos));
        }
        s.spb(pid, P_SLEFT, s.pb(pid, P_SLEFT) - 1);
        s.spb(pid, P_VP, s.pb(pid, P_VP) + 1);
        LrCache lc;                                                        // an opponent's cached longest path over this corner is gone
        lc.load(s.P);
        if (lc.settle_invalidate(pid, cn)) lc.store(s.P);
        if (flags & F_INITIAL) {
            int k = s.pb(pid, P_ISET) + 1;
            s.spb(pid, P_ISET, k);
            if (k == 2) {
                D5 d = d5_zero();
                int touched = 0;
                for (int q = 0; q < 3; q++) {
                    int t = CORNER_TILE[cn][q];
                    if (t == 255) continue;
                    int r0 = (s.b(B_TILE + t) & 15) - 1;
                    if (r0 < 0) continue;
                    s.spb(pid, P_RES + r0, s.pb(pid, P_RES + r0) + 1);
                    s.spb(pid, P_VIS + r0, s.pb(pid, P_VIS + r0) + 1);
                    s.sb(B_BANK + r0, s.b(B_BANK + r0) - 1);
                    d5_add(d, r0, 1); touched |= 1 << r0;
                }
                update_estimates(s, seatof, d, touched, pid, -1);
                s.spb(pid, P_ISECOND, cn);
            }
        } else {
            D5 d; d.v[R_BRICK] = -1; d.v[R_WOOD] = -1; d.v[R_ORE] = 0; d.v[R_SHEEP] = -1; d.v[R_WHEAT] = -1;
            update_estimates(s, seatof, d, 0b11011, pid, -1);
            int lrp = s.b(B_LR_PLAYER);
            if (lrp) {                                                     // game.py:552-553: the holder's path is looked at again
                lr_who = lrp - 1;
                // still exact in the cache (the usual case: the new settlement is not on it): nothing to search, and the
                // length equals the holder's count, so the step completes right here
                if (lc.valid(lr_who) && lr_len(lc.get(lr_who)) >= s.b(B_LR_COUNT)) { lr_inline = true; lr_len_now = lr_len(lc.get(lr_who)); }
            }
        }
        break;
    }
    case T_ROAD: {                                                         // game.py:556-597, 222-232
        bool final_init = false;
        int ed = a[2];
        if (ed != 72) {
            if (!(flags & F_INITIAL) && !(flags & F_RB_ACTIVE)) { pay(s, pid, R_WOOD, 1); pay(s, pid, R_BRICK, 1); }
            int wrow = ed < 32 ? W_ROAD0 : (ed < 64 ? W_ROAD1 : W_ROAD2);
            s.sw(wrow + pid, s.w(wrow + pid) | (1u << (ed & 31)));
            if (flags & F_INITIAL) {
                s.spb(pid, P_IROAD, s.pb(pid, P_IROAD) + 1);
                int first = 0, second = 0;
                for (int p = 0; p < 4; p++) { int k = s.pb(p, P_ISET); if (k >= 1) first++; if (k == 2) second++; }
                if (first < 4) update_players_go(s, order, false);
                else if (second == 0) { }
                else if (second < 4) update_players_go(s, order, true);
                else { flags &= ~F_INITIAL; final_init = true; }
            }
        }
        lr_who = pid;
        if (ed != 72) lr_edge = ed;
        else {                                                             // the dummy edge of road building: nothing new to search
            LrCache lc;
            lc.load(s.P);
            if (lc.valid(pid)) { lr_inline = true; lr_len_now = lr_len(lc.get(pid)); }
        }
        if (flags & F_RB_ACTIVE) {
            int k = s.b(B_RB_COUNT) + 1;
            if (k >= 2) { flags &= ~(F_RB_ACTIVE | F_MUST_USE_DEV); k = 0; }
            s.sb(B_RB_COUNT, k);
        } else if (!(flags & F_INITIAL) && !final_init) {
            D5 d = d5_zero(); d.v[R_BRICK] = -1; d.v[R_WOOD] = -1;
            update_estimates(s, seatof, d, 0b00011, pid, -1);
        }
        s.sb(B_FLAGS, flags);
        break;
    }
    case T_CITY: {                                                         // game.py:598-604, 240-251
        pay(s, pid, R_WHEAT, 2); pay(s, pid, R_ORE, 3);
        u64 bit = 1ull << a[1];
        s.set_settle(pid, s.settle(pid) & ~bit);
        s.set_city(pid, s.city(pid) | bit);
        s.spb(pid, P_VP, s.pb(pid, P_VP) + 1);
        s.spb(pid, P_CLEFT, s.pb(pid, P_CLEFT) - 1);
        s.spb(pid, P_SLEFT, s.pb(pid, P_SLEFT) + 1);
        D5 d = d5_zero(); d.v[R_ORE] = -3; d.v[R_WHEAT] = -2;
        update_estimates(s, seatof, d, (1 << R_ORE) | (1 << R_WHEAT), pid, -1);
        break;
    }
    case T_ROLL: {                                                         // game.py:605-611
        Rng rng = rng_load(c, s);
        u32 tk[3] = { 0, 0, 0 };
        int roll = roll_dice(s, rng, order, seatof, tct, cfg.prof_wave ? tk : nullptr);
        if (cfg.prof_wave != nullptr && lane == 0)          // slot 4: dice draws | tile scan + bank << 10 | hands + estimates << 20
            cfg.prof_wave[(long)wv * 8 + 4] = (tk[0] & 1023u) | ((tk[1] & 1023u) << 10) | ((tk[2] & 1023u) << 20);
        s.sw(W_RNG, rng.draws);
        flags |= F_ROLLED;
        if (roll == 7) flags |= F_CAN_ROBBER;
        s.sb(B_FLAGS, flags);
        break;
    }
    case T_ENDTURN: {                                                      // game.py:612-622
        flags &= ~(F_CAN_ROBBER | F_ROLLED | F_PLAYED_DEV);
        s.sb(B_FLAGS, flags);
        update_players_go(s, order, false);
        s.sw(W_TURN, s.w(W_TURN) + 1);
        for (int k = 0; k < 5; k++) s.sb(B_BOUGHT + k, 0);
        s.sb(B_TRADES, 0);
        s.sw(W_ACTIONS, 0);
        break;
    }
    case T_ROBBER: {                                                       // game.py:623-634
        s.sb(B_ROBBER, a[3]);
        flags &= ~F_CAN_ROBBER;
        u64 tm = topo_tile_corners(a[3]), opp = 0;
        for (int o = 0; o < 4; o++) if (o != pid) opp |= s.settle(o) | s.city(o);
        if (opp & tm) flags |= F_JUST_ROBBER;
        s.sb(B_FLAGS, flags);
        break;
    }
    case T_STEAL: {                                                        // game.py:635-652, wrapper.py:129-139
        int victim = player_at_label(order, seatof, pid, a[6]);
        int n = s.total(victim);
        if (n > 0) {
            int k;
            if (c.mt != nullptr) k = mt_choice_index(c.mt, n);             // contract (A): random.choice on the `random` module's generator
            else {
                Rng rng = rng_load(c, s);
                k = (int)rng.bounded((u32)(n - 1));
                s.sw(W_RNG, rng.draws);
            }
            const int ord[5] = { R_BRICK, R_WHEAT, R_WOOD, R_SHEEP, R_ORE };   // game.py:638
            int r0 = 0;
            bool found = false;
#pragma unroll
            for (int i = 0; i < 5; i++) {
                int h = s.res(victim, ord[i]);
                if (!found) { if (k < h) { r0 = ord[i]; found = true; } else k -= h; }
            }
            s.spb(pid, P_RES + r0, s.pb(pid, P_RES + r0) + 1);
            s.spb(victim, P_RES + r0, s.pb(victim, P_RES + r0) - 1);
            for (int q = 0; q < 5; q++) s.spb(victim, P_VIS + q, max(s.pb(victim, P_VIS + q) - 1, 0));
            D5 d = d5_zero(); d5_add(d, r0, -1);
            update_estimates(s, seatof, d, 1 << r0, victim, pid);
        }
        flags &= ~F_JUST_ROBBER;
        s.sb(B_FLAGS, flags);
        break;
    }
    case T_PLAYDEV: {                                                      // game.py:653-693, wrapper.py:140-147
        int card = a[4];
        // hidden_cards.remove(card): the ordered list lives in the COLD part of the record (global memory).  Read in batches of eight
        // bytes whose loads are all in flight together - element by element every iteration was a dependent HBM / L2 round trip
        // (0.5-1 us each with one wave per SIMD), and this type's waves were among the launch's slowest.
        int nh = s.pb(pid, P_NHID), at = -1;
        for (int base = 0; base < nh && at < 0; base += 8) {
            int v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = base + j < nh ? s.hidden(pid, base + j) : -1;
#pragma unroll
            for (int j = 0; j < 8; j++) if (at < 0 && v[j] == card) at = base + j;
        }
        if (at >= 0) {
            for (int base = at; base + 1 < nh; base += 8) {
                int v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = base + j + 1 < nh ? s.hidden(pid, base + j + 1) : 0;
#pragma unroll
                for (int j = 0; j < 8; j++) if (base + j + 1 < nh) s.set_hidden(pid, base + j, v[j]);
            }
            s.spb(pid, P_NHID, nh - 1);
            s.spb(pid, P_HCNT + card, s.pb(pid, P_HCNT + card) - 1);
        }
        int np = s.pb(pid, P_NPLAYED);
        s.set_played(pid, np, card);
        s.spb(pid, P_NPLAYED, np + 1);
        flags |= F_PLAYED_DEV;
        if (card == C_VP) s.spb(pid, P_VP, s.pb(pid, P_VP) + 1);
        else if (card == C_KNIGHT) {
            flags |= F_CAN_ROBBER;
            s.spb(pid, P_ARMY, s.pb(pid, P_ARMY) + 1);
            update_largest_army(s);
        } else if (card == C_RB) { flags |= F_RB_ACTIVE | F_MUST_USE_DEV; s.sb(B_RB_COUNT, 0); }
        else if (card == C_MONO) {
            int r0 = a[15];
            u32 lost = 0;
            int got = 0;
            for (int o = 0; o < 4; o++) if (o != pid) {
                int k = s.res(o, r0);
                s.spb(o, P_RES + r0, 0); s.spb(o, P_VIS + r0, 0);
                lost |= (u32)k << (8 * o); got += k;
            }
            s.spb(pid, P_RES + r0, s.pb(pid, P_RES + r0) + got);
            s.spb(pid, P_VIS + r0, s.pb(pid, P_VIS + r0) + got);
            update_estimates_monopoly(s, seatof, pid, r0, lost);
        } else if (card == C_YOP) {
            for (int i = 0; i < 2; i++) {
                int r0 = i == 0 ? a[15] : a[16];
                int bank = s.b(B_BANK + r0);
                if (bank > 0) {
                    s.sb(B_BANK + r0, bank - 1);
                    s.spb(pid, P_RES + r0, s.pb(pid, P_RES + r0) + 1);
                    s.spb(pid, P_VIS + r0, s.pb(pid, P_VIS + r0) + 1);
                    update_estimates1(s, seatof, r0, 1, pid);
                }
            }
        }
        s.sb(B_FLAGS, flags);
        break;
    }
    case T_BUYDEV: {                                                       // game.py:694-710
        pay(s, pid, R_SHEEP, 1); pay(s, pid, R_ORE, 1); pay(s, pid, R_WHEAT, 1);
        D5 d = d5_zero(); d.v[R_SHEEP] = -1; d.v[R_ORE] = -1; d.v[R_WHEAT] = -1;
        update_estimates(s, seatof, d, (1 << R_SHEEP) | (1 << R_ORE) | (1 << R_WHEAT), pid, -1);
        int pl = s.b(B_PILE_LEN) - 1;
        int card = s.pile(pl);
        s.sb(B_PILE_LEN, pl);
        int nh = s.pb(pid, P_NHID);
        s.set_hidden(pid, nh, card);
        s.spb(pid, P_NHID, nh + 1);
        s.spb(pid, P_HCNT + card, s.pb(pid, P_HCNT + card) + 1);
        s.sb(B_BOUGHT + card, s.b(B_BOUGHT + card) + 1);
        break;
    }
    case T_EXCHANGE: {                                                     // game.py:711-734, wrapper.py:148-153, 428-438
        int give = a[15], want = a[16];
        int hb = s.pb(pid, P_HARB);
        int rate = ((hb >> (give + 1)) & 1) ? 2 : ((hb & 1) ? 3 : 4);
        s.spb(pid, P_RES + want, s.pb(pid, P_RES + want) + 1);
        s.spb(pid, P_VIS + want, s.pb(pid, P_VIS + want) + 1);
        s.spb(pid, P_RES + give, s.pb(pid, P_RES + give) - rate);
        s.spb(pid, P_VIS + give, max(s.pb(pid, P_VIS + give) - rate, 0));
        s.sb(B_BANK + give, s.b(B_BANK + give) + rate);
        s.sb(B_BANK + want, s.b(B_BANK + want) - 1);
        D5 d = d5_zero();
        d5_add(d, want, 1); d5_add(d, give, -rate);
        update_estimates(s, seatof, d, (1 << want) | (1 << give), pid, -1);
        break;
    }
    case T_PROPOSE: {                                                      // game.py:735-750, wrapper.py:440-486
        flags |= F_MUST_RESPOND;
        s.sb(B_FLAGS, flags);
        s.sb(B_TRADE_PROP, pid);
        s.sb(B_TRADE_TGT, player_at_label(order, seatof, pid, a[6]));
        int ng = 0, nr = 0;
        bool stop = false;
        for (int i = 0; i < 4; i++) { int v = a[7 + i]; if (v <= 0 || v > 5) stop = true; s.sb(B_TRADE_GIVE + i, stop ? 0 : v); if (!stop) ng++; }
        stop = false;
        for (int i = 0; i < 4; i++) { int v = a[11 + i]; if (v <= 0 || v > 5) stop = true; s.sb(B_TRADE_RECV + i, stop ? 0 : v); if (!stop) nr++; }
        s.sb(B_TRADE_NG, ng); s.sb(B_TRADE_NR, nr);
        s.sb(B_TRADES, min(s.b(B_TRADES) + 1, 255));        // a byte: validate mode accepts proposals past the mask's limit (saturates)
        break;
    }
    case T_RESPOND: {                                                      // game.py:751-784
        if (a[5] == 0) {
            int p1 = s.b(B_TRADE_PROP), p2 = s.b(B_TRADE_TGT);
            int ng = s.b(B_TRADE_NG), nr = s.b(B_TRADE_NR);
            // The lists move one card at a time in the reference (gives first, then receives; visible counts clamp at zero on the way down).
            // Per resource that is: g cards from p1 to p2, then rc back - in closed form on the twenty hand bytes, read once and written
            // once (card by card every move was four dependent LDS read-modify-writes: up to 64 in a row with one wave per SIMD).
            int g[5] = { 0, 0, 0, 0, 0 }, rc[5] = { 0, 0, 0, 0, 0 };
            for (int i = 0; i < 4; i++) if (i < ng) {
                const int r0 = s.b(B_TRADE_GIVE + i) - 1;
#pragma unroll
                for (int k = 0; k < 5; k++) g[k] += (k == r0) ? 1 : 0;
            }
            for (int i = 0; i < 4; i++) if (i < nr) {
                const int r0 = s.b(B_TRADE_RECV + i) - 1;
#pragma unroll
                for (int k = 0; k < 5; k++) rc[k] += (k == r0) ? 1 : 0;
            }
            D5 d1, d2;
            int t1 = 0;
            int r1[5], v1[5], r2[5], v2[5];
#pragma unroll
            for (int k = 0; k < 5; k++) { r1[k] = s.pb(p1, P_RES + k); v1[k] = s.pb(p1, P_VIS + k); r2[k] = s.pb(p2, P_RES + k); v2[k] = s.pb(p2, P_VIS + k); }
#pragma unroll
            for (int k = 0; k < 5; k++) {
                s.spb(p1, P_RES + k, r1[k] - g[k] + rc[k]);
                s.spb(p1, P_VIS + k, max(v1[k] - g[k], 0) + rc[k]);
                s.spb(p2, P_RES + k, r2[k] + g[k] - rc[k]);
                s.spb(p2, P_VIS + k, max(v2[k] + g[k] - rc[k], 0));
                d1.v[k] = rc[k] - g[k]; d2.v[k] = g[k] - rc[k];
                t1 |= (g[k] | rc[k]) ? 1 << k : 0;
            }
            update_estimates(s, seatof, d1, t1, p1, -1);
            update_estimates(s, seatof, d2, t1, p2, -1);
        }
        flags &= ~F_MUST_RESPOND;
        s.sb(B_FLAGS, flags);
        s.sb(B_TRADE_PROP, 0); s.sb(B_TRADE_TGT, 0); s.sb(B_TRADE_NG, 0); s.sb(B_TRADE_NR, 0);
        for (int i = 0; i < 4; i++) { s.sb(B_TRADE_GIVE + i, 0); s.sb(B_TRADE_RECV + i, 0); }
        break;
    }
    case T_DISCARD: {                                                      // game.py:785-807
        int who = s.b(B_DISC), r0 = a[17];
        s.spb(who, P_RES + r0, s.pb(who, P_RES + r0) - 1);
        s.sb(B_BANK + r0, s.b(B_BANK + r0) + 1);
        update_estimates1(s, seatof, r0, -1, who);
        if (s.total(who) <= 7) {
            int n = s.b(B_NDISC);
            for (int i = 0; i + 1 < 4; i++) s.sb(B_DISC + i, (i + 1 < n) ? s.b(B_DISC + i + 1) : 0);
            s.sb(B_DISC + 3, 0);
            s.sb(B_NDISC, n - 1);
        }
        break;
    }
    default: break;
    }
    if (cfg.prof_wave != nullptr) {                         // slot 3: before the switch (validate, clamps) | the switch << 16; slot 4: unused
        const long long t_sw1 = clock_fenced();
        if (lane == 0) cfg.prof_wave[(long)wv * 8 + 3] = ((u32)(t_sw0 - tprof) & 0xFFFFu) | ((u32)(t_sw1 - t_sw0) << 16);
    }
    if (type >= 0 && type != T_RESPOND && type != T_ENDTURN && type != T_DISCARD) s.sw(W_ACTIONS, s.w(W_ACTIONS) + 1);   // game.py:809-810

    if (cfg.prof_wave != nullptr && lane == 0) cfg.prof_wave[(long)wv * 8 + 5] = (u32)(type >= 0 ? bin + 1 : 0);   // slot 5: sort bin + 1
    if (cfg.prof != nullptr && cfg.prof_wave == nullptr && lane == 0) {   // per action type: time of validate+apply
        const int t0 = live ? actions[e * ACTION_WORDS] : 13;
        const int tb = (t0 < 0 || t0 > 12) ? 13 : t0;
        const unsigned long long dt = (unsigned long long)(wall_clock64() - tprof);
        atomicAdd(&cfg.prof[PROF_TOTAL + tb], dt); atomicAdd(&cfg.prof[PROF_TOTAL + 14 + tb], 1ull);
        atomicMax(&cfg.prof[PROF_TOTAL + 28 + tb], dt);
    }
    prof_mark(cfg, 1, tprof);
    // ---- update_longest_road (game.py:864-919): the path search runs in k_lr_finish / k_lr_heavy, which also complete
    // the step of these games (sorted waves would otherwise serialise up to 64 searches in the road-placement waves)
    const int len = lr_len_now;
    const bool pending = lr_who >= 0 && !lr_inline;
    if (pending) {
        const u32 slot = atomicAdd(&pend.ctr[lrq_ctr(pend.fa)], 1u);
        pend.req[pend.fa][slot] = (u64)e | ((u64)lr_edge << 40) | ((u64)lr_who << 56);
        if (pend.stag < 2) {                               // lock-step: a game whose completion can end it (longest road: +2 points for
            bool may_end = false;                          // one player) gets a speculative successor (k_reset_list)
            if (type == T_ROAD) {
                // a new road can only hand the longest road to its builder (+2 for him, -2 for the previous holder): the game can
                // end only if he has 8 points and does not hold it already.  (A superset filter is all that is needed -
                // k_install_list re-deals a finished game without a shadow - but every speculative re-deal lengthens
                // k_reset_list, whose duration is its slowest re-deal.)
                may_end = s.pb(lr_who, P_VP) >= 8 && s.b(B_LR_PLAYER) != lr_who + 1;
            } else {
#pragma unroll
                for (int p = 0; p < 4; p++) may_end |= s.pb(p, P_VP) >= 8;
            }
            if (may_end) pend.spec[atomicAdd(&pend.ctr[6], 1u)] = (u64)e;
        }
        pend.type[e] = (u8)(type + 1);
        pend.who[e] = (u8)lr_who;
        pend.busy[e] = (u8)pend.ftag;
    }
    prof_mark(cfg, 2, tprof);
    u32 m_new[MASK_WORDS];
    bool have_masks = false;
    finish_step<1>(c, s, (StepScratch*)nullptr, cfg, lane, type >= 0 && !pending, type, lr_inline ? lr_who : -1, len, reward, done, mpk, tprof, nbr_c, nbr_e,
                   pend, 0, false, m_new, &have_masks);
    // ---- SAMPLE: the next action of every game completed here (random policy, decision index = the advanced counter), and
    // the game's place in the next pass's lists: per bin one atomic by the first lane that drew it, ranks from the ballots
    int an[ACTION_WORDS] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    u32 abase = 0, arank = 0;
    int aleader = 0, anb = -1;
    if constexpr (SAMPLE) {
        const long long t_s0 = cfg.prof_wave ? clock_fenced() : 0;
        if (type >= 0) dctr++;                              // the action was applied: one more decision of this game
        int nb = -1;
        if (have_masks) nb = bin_of(sample_random(c, s, m_new, dctr, an, (u32*)nullptr, (u8*)nullptr, 0, 0), an[4]);
        else if (type >= 0) mpk[e * MPK_STRIDE + ROW_CTR] = dctr;    // (slow path / re-deal: the completing kernel draws with this counter)
        const long long t_s1 = cfg.prof_wave ? clock_fenced() : 0;
        u32 rank = 0, cntb = 0;
        int leader = 0;
#pragma unroll
        for (int b = 0; b < NBINS - 1; b++) {
            const u64 mk = __ballot(nb == b);
            if (nb == b) { rank = (u32)__popcll(mk & ((1ull << lane) - 1)); cntb = (u32)__popcll(mk); leader = __ffsll((long long)mk) - 1; }
        }
        // (the atomic's round trip - device scope: ~2 us - runs under the write-back of the records below; its result is used after it)
        if (nb >= 0 && lane == leader) abase = atomicAdd(sub_ctr(pend, pend.bnext, nb, wv & (pend.nsub - 1)), cntb);
        arank = rank; aleader = leader; anb = nb;
        if (cfg.prof_wave != nullptr) {                     // slot 4 (SAMPLE): the draw | the ranking << 16
            const long long t_s2 = clock_fenced();
            if (lane == 0) cfg.prof_wave[(long)wv * 8 + 4] = ((u32)(t_s1 - t_s0) & 0xFFFFu) | ((u32)(t_s2 - t_s1) << 16);
        }
    }
    // ---- write the tile back, then the new mask rows (SAMPLE: the whole side rows) through the tile
    __builtin_amdgcn_wave_barrier();
    if (w_board) stage_out<28, 0, G>(tile, c.R, (type >= 0 || rejected) ? (int)e : -1, lane);
    else if (w_est) stage_out<21, 7, G>(tile, c.R, (type >= 0 || rejected) ? (int)e : -1, lane);
    else stage_out<12, 16, G>(tile, c.R, (type >= 0 || rejected) ? (int)e : -1, lane);
    __builtin_amdgcn_wave_barrier();
    if constexpr (SAMPLE) {
        abase = __shfl(abase, aleader);
        if (anb >= 0) sub_list(pend, pend.bnext, anb, wv & (pend.nsub - 1), c.N)[abase + arank] = (i32)e;
    }
    if constexpr (SAMPLE) stage_out_row<G>(tile, mpk, have_masks ? (int)e : -1, lane, m_new, an, dctr);
    else stage_out_masks<G>(tile, mpk, have_masks ? (int)e : -1, lane, m_new);
    prof_mark(cfg, 7, tprof);
    if (cfg.prof_wave != nullptr && cfg.prof_timeline) {
        u32 hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (lane == 0) cfg.prof_wave[(long)wv * 8 + 3] = (hw & 0x0FFFFFFFu) | ((xcc & 15u) << 28);
    }
}

constexpr u64 LR_GAME_MASK = (1ull << 40) - 1;             // request: game | new edge (127: none) << 40 | pid0 << 56
// What a longest-road request needs: THROUGH(new edge) when the player's cached path is exact, else FULL.  (A request of
// a settlement is always FULL: k_step completes those whose holder's cache is intact itself.)
struct LrPlan { bool through; int u, v; };
DEVI LrPlan lr_plan(const LrCache& lc, int who, int edge) {
    LrPlan pl;
    pl.through = edge < 72 && lc.valid(who);
    const int ed = edge < 72 ? edge : 0;
    pl.u = EDGE_CORNER[ed][0]; pl.v = EDGE_CORNER[ed][1];
    return pl;
}
// the player's new longest path from a search result: length and the cache entry (valid)
DEVI int lr_apply(LrCache& lc, int who, bool through, u64 found) {
    const u64 old = lc.get(who);
    const int flen = (int)(found >> 54);
    if (through && lr_len(old) >= flen) return lr_len(old);
    lc.set(who, found & LR_VMASK);
    return flen;
}
// Tier 1 of the longest road and the completion of the step, one request per wave: all 64 lanes cooperate on the path
// search (budgeted; overflow hands the game to the tier-2 list), then the game's hot record is staged linearly in LDS and
// lane 0 completes the step (holder logic, done/reward, next masks).  `fl` selects the request list.
// LRF_SPLIT (the deferred schedules' tier 1; CATAN_LR_SPLIT): the wave only searches.  Unless the holder's own road was cut (rare: finish_step<2>
// then needs the other players' paths, possibly more searches) it stores the cache, leaves the new length in pend.len[game] with bit 63
// set and goes on to its next request; k_lr_complete, launched behind this kernel, completes those steps LANE per game - the one-lane
// completion (holder logic, done / rewards, compute_masks by lane 0: 3.7 of a request's 8.5 us) was nearly half of tier 1's wave time,
// and tier 1's waves share the SIMDs with the next pass's sampler and k_step.
// LRF_MID (the MIDDLE tier of a deferred window; CATAN_LR_MID_BUDGET): the same one-wave search with a much larger
// budget over the window's TIER-2 requests (pend.heavy[sa]) in front of k_lr_heavy, completing what it finishes as the tier-2 completion
// would (re-deal list 1, window tag) and handing only what still overflows to k_lr_heavy (pend.heavy2, ctr[CTR_HEAVY2]).  A tier-2 workgroup
// is 1 024 threads with 142 KB of LDS and every register of its CU: while k_lr_heavy runs (606 us of a 1.7 ms window on 128 CUs) half the
// machine takes no k_step wave - profiles/r05_dispatch_ramp.txt - and most of its requests are searches one wave ends in a few hundred
// iterations.
constexpr int CTR_HEAVY2 = 88;
constexpr int LRF_TIER1 = 0, LRF_SPLIT = 1, LRF_MID = 2;
template <int MODE>
__global__ __launch_bounds__(64) void k_lr_finish(Ctx c, u32* __restrict__ mpk, float* __restrict__ reward, u8* __restrict__ done,
                                                  StepCfg cfg, Pending pend, int fl, int budget, unsigned long long* stat,
                                                  unsigned long long* slow_ctr) {
    __shared__ StepScratch scratch;
    __shared__ __attribute__((aligned(16))) u32 rec[ROWS_HOT];
    const int lane = threadIdx.x;
    u32 nbr_c, nbr_e;
    lr_load_nbr(lane, nbr_c, nbr_e);
    constexpr bool SPLIT = MODE == LRF_SPLIT, MID = MODE == LRF_MID;
    const u32 count = MID ? pend.ctr[8 + 4 * pend.sa] : pend.ctr[lrq_ctr(fl)];
    if (!MID && blockIdx.x == 0 && lane == 0 && slow_ctr != nullptr) { atomicAdd(&slow_ctr[0], (unsigned long long)count); atomicAdd(&slow_ctr[2], 1ull); }
    StepCfg cfg2 = cfg;
    cfg2.prof = nullptr;
    // per-request phase ticks (catan_profile_enable(env, 2)): rows LRF_PROF_ROW.. of the per-wave buffer, one request per workgroup
    cfg2.prof_wave = cfg.prof_wave != nullptr && blockIdx.x < LRF_PROF_ROWS && count <= gridDim.x && c.N >= 65536 ? cfg.prof_wave + LRF_PROF_ROW * 8 : nullptr;
    for (u32 r = blockIdx.x; r < count; r += gridDim.x) {
        long long tprof = wall_clock64();
        const u64 rq = MID ? pend.heavy[pend.sa][r] : pend.req[fl][r];
        const long e = (long)(rq & LR_GAME_MASK);
        const int who = (int)(rq >> 56), edge = (int)((rq >> 40) & 127);
        __builtin_amdgcn_wave_barrier();
        if (lane < ROWS_HOT / 4) reinterpret_cast<uint4*>(rec)[lane] = reinterpret_cast<const uint4*>(c.R + e * REC)[lane];
        __builtin_amdgcn_wave_barrier();
        prof_mark(cfg2, 0, tprof);
        StL1 s(rec, c.R, c.N, e);
        LrCache lc;
        lc.load(s.P);                                     // (every lane: the same seven words)
        const LrPlan pl = lr_plan(lc, who, edge);
        u64 BL = 0;
        for (int o = 0; o < 4; o++) if (o != who) BL |= s.settle(o) | s.city(o);
        const u64 found = lr_wave_search(s.road_lo(who), s.road_hi(who), BL, pl.through, pl.u, pl.v, scratch.lr, budget, nbr_c, nbr_e, stat);
        if (MID && found == LR_OVERFLOW) {               // still too deep for one wave: k_lr_heavy (len / arrive / busy were set when tier 1 gave up)
            if (lane == 0) pend.heavy2[atomicAdd(&pend.ctr[CTR_HEAVY2], 1u)] = rq;
            continue;
        }
        if (found == LR_OVERFLOW) {                      // tier 2 takes over; the record is untouched
            if (lane == 0) {
                const u32 slot = atomicAdd(&pend.ctr[8 + 4 * pend.sa], 1u);
                pend.heavy[pend.sa][slot] = rq; pend.len[e] = 0; pend.arrive[e] = 0; pend.busy[e] = (u8)pend.stag;
                if (slow_ctr != nullptr) atomicAdd(&slow_ctr[1], 1ull);
            }
            continue;
        }
        const int len = lr_apply(lc, who, pl.through, found);
        prof_mark(cfg2, 1, tprof);
        if constexpr (SPLIT) {
            if (!(s.b(B_LR_PLAYER) == who + 1 && s.b(B_LR_COUNT) > len)) {      // (wave-uniform: every lane reads the same record)
                if (lane == 0) { lc.store(s.P); pend.len[e] = (1ull << 63) | (u64)len; }
                continue;
            }
            if (lane == 0) pend.len[e] = 0;               // the cut case is completed here: k_lr_complete must find no mark (whatever the word held)
        }
        u32 mn[MASK_WORDS];
        bool mv = false;
        finish_step<2>(c, s, &scratch, cfg2, lane, lane == 0, (int)pend.type[e] - 1, who, len, reward, done, mpk, tprof, nbr_c, nbr_e, pend,
                       MID ? 1 : (pend.stag < 2 ? 2 : 0), MID ? pend.stag < 2 : pend.ftag < 2, pend.sample ? mn : nullptr, pend.sample ? &mv : nullptr, &lc);
        if (mv) {                                         // fused-sampling rollouts (lane 0, the game goes on): masks, next action, its place in the pass it returns in
#pragma unroll
            for (int i = 0; i < MASK_WORDS; i++) mpk[e * MPK_STRIDE + i] = mn[i];
            sample_enqueue_lane(c, s, mn, pend, mpk);
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < ROWS_HOT / 4) reinterpret_cast<uint4*>(c.R + e * REC)[lane] = reinterpret_cast<const uint4*>(rec)[lane];
        if (lane == 0) lc.store(s.P);
        prof_mark(cfg2, 7, tprof);
    }
}

// The completion of the tier-1 requests k_lr_finish<true> left (pend.len[game] bit 63): lane per game through an LDS tile, as k_step
// completes the games whose length the cache knew - holder logic (no cut among these), done / rewards, next masks, write-back.
__global__ __launch_bounds__(64) void k_lr_complete(Ctx c, u32* __restrict__ mpk, float* __restrict__ reward, u8* __restrict__ done,
                                                    StepCfg cfg, Pending pend, int fl) {
    __shared__ u32 tile[ROWS_HOT * TS];
    const int lane = threadIdx.x;
    const u32 count = pend.ctr[lrq_ctr(fl)];
    cfg.prof = nullptr; cfg.prof_wave = nullptr;
    for (u32 base = blockIdx.x * 64u; base < count; base += gridDim.x * 64u) {
        const u32 r = base + lane;
        long e = -1;
        int who = 0, len = 0, type = -1;
        if (r < count) {
            const u64 rq = pend.req[fl][r];
            const long g = (long)(rq & LR_GAME_MASK);
            const u64 L = pend.len[g];
            if (L >> 63) { e = g; who = (int)(rq >> 56); len = (int)(L & 0xFFu); type = (int)pend.type[g] - 1; pend.len[g] = 0; }
        }
        if (__ballot(e >= 0) == 0) continue;
        __builtin_amdgcn_wave_barrier();
        stage_in(tile, c.R, (int)e, lane);
        __builtin_amdgcn_wave_barrier();
        StL s(tile + lane, c.R, c.N, e >= 0 ? e : 0);
        u32 m_new[MASK_WORDS];
        bool have_masks = false;
        long long tprof = 0;
        finish_step<1>(c, s, (StepScratch*)nullptr, cfg, lane, e >= 0, type, who, len, reward, done, mpk, tprof, 0u, 0u, pend,
                       pend.stag < 2 ? 2 : 0, pend.ftag < 2, m_new, &have_masks);
        // fused-sampling rollouts: the next action of every game completed here (decision index = the counter k_step left in the side row) and its
        // place in the pass it returns in - as sample_enqueue_lane does for the one-lane completions, here lane per game
        int an[ACTION_WORDS] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        u32 dctr = 0;
        if (pend.sample && have_masks) {
            dctr = mpk[e * MPK_STRIDE + ROW_CTR];
            const int nb = bin_of(sample_random(c, s, m_new, dctr, an, (u32*)nullptr, (u8*)nullptr, 0, 0), an[4]), sub = (int)(e & (pend.nsub - 1));
            sub_list(pend, pend.brel, nb, sub, c.N)[atomicAdd(sub_ctr(pend, pend.brel, nb, sub), 1u)] = (i32)e;
        }
        __builtin_amdgcn_wave_barrier();
        stage_out<28, 0, 64>(tile, c.R, (int)e, lane);
        __builtin_amdgcn_wave_barrier();
        if (pend.sample) stage_out_row<64>(tile, mpk, have_masks ? (int)e : -1, lane, m_new, an, dctr);
        else stage_out_masks<64>(tile, mpk, have_masks ? (int)e : -1, lane, m_new);
    }
}

// tier 2: `split` workgroups per request (FULL: static partition of the start corners; THROUGH: the single seed starts
// in part 0, the other parts only arrive; results combined with a 64-bit atomicMax of len << 54 | vertex mask); few requests
// (lock-step) get 8 workgroups each for latency, many (a deferred window) share the grid for throughput.
// req[i] = game | new edge << 40 | pid0 << 56; out64[game] (zeroed when the request was pushed) collects the result.  The
// workgroup that arrives LAST at a request (pend.arrive[game]) completes the step of that game - cache, holder logic,
// done / rewards, next masks - on its first wave, as k_lr_finish does for tier 1 (a separate completion kernel cost 28 us of
// every lock-step step).
__global__ __launch_bounds__(LR_HEAVY_THREADS) void k_lr_heavy(Ctx c, const u32* __restrict__ req_count,
                                                              const u64* __restrict__ req, u64* __restrict__ out64, int round_iters,
                                                              u32* __restrict__ mpk, float* __restrict__ reward, u8* __restrict__ done,
                                                              StepCfg cfg, Pending pend) {
    __shared__ u64 adj[54];
    __shared__ u64 pool_seen[LR_POOL];
    __shared__ unsigned short pool_cd[LR_POOL];
    __shared__ unsigned long long best_all;
    __shared__ u64 blk_all;
    __shared__ int pool_n, is_last;
    __shared__ __attribute__((aligned(16))) lrstk_t path[54][LR_HEAVY_THREADS];
    static_assert(sizeof(StepScratch) + ROWS_HOT * 4 + 64 <= sizeof(lrstk_t) * 54 * LR_HEAVY_THREADS, "the completion reuses the DFS stacks");
    const int tid = threadIdx.x;
    const u32 nreq = *req_count;
    const u32 LR_SPLIT = nreq * 8 <= gridDim.x ? 8u : (nreq * 4 <= gridDim.x ? 4u : (nreq * 2 <= gridDim.x ? 2u : 1u));
    const u32 count = nreq * LR_SPLIT;
    u32 nbr_c, nbr_e;
    lr_load_nbr(tid < 54 ? tid : 0, nbr_c, nbr_e);
    StepCfg cfg2 = cfg;
    cfg2.prof = nullptr; cfg2.prof_wave = nullptr;
    // per-workgroup profile of its first request (catan_profile_enable(env, 2)): [0] set-up ticks, [1] rounds, [2] search ticks,
    // [3] combine + arrival ticks, [4] completion ticks (the last part), [5] split | through << 8 | part << 16, [6] / [7] clock at start / end
    u32* const hp = cfg.prof_wave != nullptr && blockIdx.x < LRH_PROF_ROWS && c.N >= 65536 ? cfg.prof_wave + (LRH_PROF_ROW + blockIdx.x) * 8 : nullptr;
    for (u32 r = blockIdx.x; r < count; r += gridDim.x) {
        const long long hp_t0 = wall_clock64();
        u32 hp_rounds = 0;
        const u64 rq = req[r / LR_SPLIT];
        const int part = (int)(r % LR_SPLIT);
        const long game = (long)(rq & LR_GAME_MASK);
        const int pid = (int)(rq >> 56), edge = (int)((rq >> 40) & 127);
        St s(c.R, c.N, game);
        LrCache lc;
        lc.load(s.P);                              // (no part writes the cache before every part has arrived)
        const LrPlan pl = lr_plan(lc, pid, edge);
        __syncthreads();
        if (tid < 54) {
            u64 BL = 0;
            for (int o = 0; o < 4; o++) if (o != pid) BL |= s.settle(o) | s.city(o);
            adj[tid] = lr_adj_of(tid, nbr_c, nbr_e, s.road_lo(pid), s.road_hi(pid), BL);
            if (tid == 0) blk_all = BL;
        }
        if (tid == 0) { pool_n = 0; best_all = 0ull; }
        __syncthreads();
        const LrGraph G{ adj, blk_all, pl.v };
        Dfs t;
        if (pl.through) {                                  // 4^5 prefixes = the 1 024 threads of part 0; with >= 4 parts: 4^6 over parts 0..3
            const bool wide = LR_SPLIT >= 4;
            dfs_seed_through(t, wide ? part < 4 : part == 0, pl.u, pl.v, adj[pl.u]);
            dfs_walk_prefix<true>(t, G, wide ? part * LR_HEAVY_THREADS + tid : tid, wide ? 6 : 5);
        } else {                                           // start corner tid % 54 (this part's share), 16 two-level prefixes each
            const int st = tid % 54, code = tid / 54;
            dfs_seed_full(t, st, adj[st]);
            t.active = t.active && code < 16 && ((u32)st % LR_SPLIT) == (u32)part;
            dfs_walk_prefix<true>(t, G, code, 2);
        }
        const DfsQueue q{ pool_seen, pool_cd, &pool_n, LR_POOL };
        bool hint = true;
        const long long hp_t1 = wall_clock64();
        while (true) {
            hp_rounds++;
            for (int it = 0; it < round_iters; it++) dfs_iter<true>(t, G, &path[0][tid], LR_HEAVY_THREADS, hint, q);
            __syncthreads();                       // all pushes of this round are complete
            if (!t.active) {
                const int qi = atomicSub(&pool_n, 1) - 1;
                if (qi >= 0) dfs_take(t, G, pool_seen[qi], pool_cd[qi]);
                else atomicAdd(&pool_n, 1);
            }
            const int busy = __syncthreads_count(t.active);   // all pops complete
            if (busy == 0) break;                  // nobody active -> the pool is empty too (idle threads drained it)
            hint = busy < LR_HEAVY_THREADS;
        }
        const long long hp_t2 = wall_clock64();
        if (t.best > 0) atomicMax(&best_all, (unsigned long long)lr_pack(t.best, t.bseen));
        __syncthreads();
        if (tid == 0) {
            if (best_all > 0) atomicMax(reinterpret_cast<unsigned long long*>(&out64[game]), best_all);
            __threadfence();                                           // this part's result before its arrival
            is_last = atomicAdd(&pend.arrive[game], 1u) == LR_SPLIT - 1 ? 1 : 0;
        }
        __syncthreads();
        const long long hp_t3 = wall_clock64();
        if (is_last && tid < 64) {                                     // every part has arrived: complete the step of this game
            StepScratch* scratch = reinterpret_cast<StepScratch*>(&path[0][0]);
            u32* rec = reinterpret_cast<u32*>(reinterpret_cast<char*>(&path[0][0]) + ((sizeof(StepScratch) + 63) & ~size_t(63)));
            const int lane = tid;
            const u64 found = atomicMax(reinterpret_cast<unsigned long long*>(&out64[game]), 0ull);   // (device-scope read of the combined result)
            if (lane < ROWS_HOT / 4) reinterpret_cast<uint4*>(rec)[lane] = reinterpret_cast<const uint4*>(c.R + game * REC)[lane];
            __builtin_amdgcn_wave_barrier();
            StL1 sl(rec, c.R, c.N, game);
            u32 nc, ne;
            lr_load_nbr(lane, nc, ne);
            const int len = lr_apply(lc, pid, pl.through, found);
            long long tprof = 0;
            u32 mn[MASK_WORDS];
            bool mv = false;
            finish_step<2>(c, sl, scratch, cfg2, lane, lane == 0, (int)pend.type[game] - 1, pid, len, reward, done, mpk, tprof, nc, ne, pend,
                           1, pend.stag < 2, pend.sample ? mn : nullptr, pend.sample ? &mv : nullptr, &lc);
            if (mv) {                                     // fused-sampling rollouts: as in k_lr_finish (pend.brel < 0: the window's release list)
#pragma unroll
                for (int i = 0; i < MASK_WORDS; i++) mpk[game * MPK_STRIDE + i] = mn[i];
                sample_enqueue_lane(c, sl, mn, pend, mpk);
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < ROWS_HOT / 4) reinterpret_cast<uint4*>(c.R + game * REC)[lane] = reinterpret_cast<const uint4*>(rec)[lane];
            if (lane == 0) lc.store(sl.P);
        }
        if (hp != nullptr && tid == 0 && r == blockIdx.x) {
            const long long hp_t4 = wall_clock64();
            hp[0] = (u32)(hp_t1 - hp_t0); hp[1] = hp_rounds; hp[2] = (u32)(hp_t2 - hp_t1); hp[3] = (u32)(hp_t3 - hp_t2);
            hp[4] = is_last ? (u32)(hp_t4 - hp_t3) : 0u; hp[5] = LR_SPLIT | ((pl.through ? 1u : 0u) << 8) | ((u32)part << 16);
            hp[6] = (u32)hp_t0; hp[7] = (u32)hp_t4;
        }
    }
}

// Deferred rollouts: frees the games that still carry a release tag when a call ends (their steps are complete).
__global__ __launch_bounds__(BLOCK) void k_release_tags(Ctx c, u8* __restrict__ busy) {
    const long e = (long)blockIdx.x * BLOCK + threadIdx.x;
    if (e < c.N && busy[e] >= 2) busy[e] = 0;
}

// ---- fused-sampling deferred rollouts: the three small kernels around the loop
// first pass of a call: every game's first action (decision index pctr[e], which is NOT advanced: k_step counts a decision when
// it applies it) into its side row, the counter next to it, the game into bin set 0
__global__ __launch_bounds__(BLOCK) void k_sample_first(Ctx c, u32* __restrict__ mpk, const u32* __restrict__ pctr, Pending pend, int set) {
    __shared__ u32 hist[NBINS], base[NBINS];
    if (threadIdx.x < NBINS) hist[threadIdx.x] = 0;
    __syncthreads();
    St s(c.R, c.N, (long)blockIdx.x * BLOCK + threadIdx.x);
    int bin = BIN_NOOP;
    if (s.e < c.n) {
        u32* row = mpk + s.e * MPK_STRIDE;
        u32 m[MASK_WORDS];
#pragma unroll
        for (int i = 0; i < MASK_WORDS; i++) m[i] = row[i];
        int a[ACTION_WORDS];
        const u32 d = pctr[s.e];
        bin = bin_of(sample_random(c, s, m, d, a, (u32*)nullptr, (u8*)nullptr, 0, 0), a[4]);
        uint4* dst = reinterpret_cast<uint4*>(row + ROW_ACT);
#pragma unroll
        for (int i = 0; i < 4; i++) dst[i] = make_uint4((u32)a[4 * i], (u32)a[4 * i + 1], (u32)a[4 * i + 2], (u32)a[4 * i + 3]);
        dst[4] = make_uint4((u32)a[16], (u32)a[17], d, 0u);
    }
    // (sort_append, defined further down, inlined: the block's games appended to the per-bin lists)
    u32 rank = 0;
    const bool valid = s.e < c.n;
    if (valid) rank = atomicAdd(&hist[bin], 1u);
    __syncthreads();
    const int sub = (int)(blockIdx.x & (pend.nsub - 1));
    if (threadIdx.x < NBINS) base[threadIdx.x] = hist[threadIdx.x] ? atomicAdd(sub_ctr(pend, set, threadIdx.x, sub), hist[threadIdx.x]) : 0u;
    __syncthreads();
    if (valid) sub_list(pend, set, bin, sub, c.N)[base[bin] + rank] = (i32)s.e;
}
// opening of window w + 2: the games whose step the slow path of window w completed (tier 2, re-deals) sit in that window's
// release list with their next action already in their side rows - enqueue them for this pass
__global__ __launch_bounds__(BLOCK) void k_release_window(Ctx c, const u32* __restrict__ mpk, const u32* __restrict__ count_p, const i32* __restrict__ list,
                                                         Pending pend, int set) {
    const u32 count = *count_p;
    for (u32 r = blockIdx.x * BLOCK + threadIdx.x; r < count; r += gridDim.x * BLOCK) {
        const long e = list[r];
        const u32* row = mpk + e * MPK_STRIDE;
        const int bin = bin_of((int)row[ROW_ACT], (int)row[ROW_ACT + 4]), sub = (int)(e & (pend.nsub - 1));
        sub_list(pend, set, bin, sub, c.N)[atomicAdd(sub_ctr(pend, set, bin, sub), 1u)] = (i32)e;
    }
}
// end of a call: the decision counters back into the handle's array (catan_policy_counters), every slow-path tag cleared
__global__ __launch_bounds__(BLOCK) void k_finish_rollout(Ctx c, const u32* __restrict__ mpk, u32* __restrict__ pctr, u8* __restrict__ busy) {
    const long e = (long)blockIdx.x * BLOCK + threadIdx.x;
    if (e < c.n) pctr[e] = mpk[e * MPK_STRIDE + ROW_CTR];
    if (e < c.N) busy[e] = 0;
}

// One wave resets one game: the 64 lanes generate the game's next RND_WORDS Philox draws into LDS, lane 0 runs the
// (inherently serial) shuffles of Board.reset / Game.reset on the game's hot record held linearly in LDS, then computes
// the masks of the fresh game.
// dstR != nullptr: a SPECULATIVE re-deal - the fresh record (and its masks, through `mpk`) go to the shadow arrays, the game
// itself is only read (its stream position).  k_install_list copies the shadow over the game if the game did end.
DEVI void wave_reset_game(const Ctx& c, long e, u32* rec, ResetScratch& sc, int lane, u32* __restrict__ mpk, Limits lim, u8* busy,
                           unsigned long long* prof = nullptr, u32* __restrict__ dstR = nullptr, const Pending* pend = nullptr, bool board_only = false) {
    u32* const outR = dstR ? dstR : c.R;
    __builtin_amdgcn_wave_barrier();
    if (lane < ROWS_HOT / 4) reinterpret_cast<uint4*>(rec)[lane] = reinterpret_cast<const uint4*>(c.R + e * REC)[lane];
    __builtin_amdgcn_wave_barrier();
    StL1 s(rec, outR, c.N, e);
    Rng mine = rng_load(c, s);
    const u32 blk0 = mine.draws >> 2;
#pragma unroll
    for (int b = 0; b < RND_WORDS / 256; b++) {
        u32 o[4];
        mine.block(blk0 + b * 64 + lane, o);
        const int at = (b * 64 + lane) * 4;
        sc.rnd[at] = o[0]; sc.rnd[at + 1] = o[1]; sc.rnd[at + 2] = o[2]; sc.rnd[at + 3] = o[3];
    }
    for (int r = ROWS_HOT + lane; r < REC; r += 64) outR[e * REC + r] = 0;    // cold part: empty card lists
    __builtin_amdgcn_wave_barrier();
    RngBuf rb;
    rb.buf = sc.rnd; rb.base = blk0 * 4; rb.avail = RND_WORDS; rb.slow = mine;
    const long long t0 = prof ? wall_clock64() : 0;
    if (lane == 0) reset_part1(s, rb, sc, ROWS_HOT);
    __builtin_amdgcn_wave_barrier();
    const bool tokens_done = reset_tokens_parallel(sc, lane);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        reset_part2(s, rb, sc, tokens_done, board_only);
        if (prof) {
            const unsigned long long dt = (unsigned long long)(wall_clock64() - t0);
            atomicAdd(&prof[5], dt); atomicMax(&prof[PROF_PHASES + 5], dt);
            atomicAdd(&prof[3], (unsigned long long)(rb.slow.draws - mine.draws)); atomicMax(&prof[PROF_PHASES + 3], (unsigned long long)(rb.slow.draws - mine.draws));
            atomicAdd(&prof[4], 1ull);
        }
        if (mpk != nullptr) {
            u32 m[MASK_WORDS];
            compute_masks(s, m, lim);
#pragma unroll
            for (int i = 0; i < MASK_WORDS; i++) mpk[e * MPK_STRIDE + i] = m[i];
            if (pend != nullptr && pend->sample) sample_enqueue_lane(c, s, m, *pend, mpk);   // fused-sampling rollouts: the fresh game's first action
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < ROWS_HOT / 4) reinterpret_cast<uint4*>(outR + e * REC)[lane] = reinterpret_cast<const uint4*>(rec)[lane];
    if (lane == 0 && busy != nullptr) busy[e] = 0;
}
// Resets the games this step finished (RL/ppo/game_manager.py:112-113), one wave per game.
// spec_list != nullptr (lock-step steps): the launch also deals, speculatively and into the shadow arrays, a fresh game for
// every game on the tier-2 list - if such a game turns out to end, k_install_list copies the shadow over it
// instead of a second, fully exposed re-deal pass at the end of the step (a re-deal depends only on the game's stream
// position, which the rest of the step does not move).
__global__ __launch_bounds__(64) void k_reset_list(Ctx c, u32* __restrict__ mpk, Limits lim, const u32* __restrict__ count_p,
                                                   const i32* __restrict__ list, u8* __restrict__ busy, unsigned long long* prof,
                                                   const u32* __restrict__ spec_count_p, const u64* __restrict__ spec_list,
                                                   u32* __restrict__ specR, u32* __restrict__ spec_mpk, u32 epoch, Pending pend) {
    __shared__ __attribute__((aligned(16))) u32 rec[ROWS_HOT];
    __shared__ ResetScratch sc;
    const u32 count = *count_p, scount = spec_list ? *spec_count_p : 0u;
    for (u32 r = blockIdx.x; r < count + scount; r += gridDim.x) {
        if (r < count) wave_reset_game(c, list[r], rec, sc, threadIdx.x, mpk, lim, busy, prof, nullptr, &pend);
        else {
            const long e = (long)(spec_list[r - count] & 0x00FFFFFFFFFFFFFFull);
            wave_reset_game(c, e, rec, sc, threadIdx.x, spec_mpk, lim, nullptr, nullptr, specR);
            if (threadIdx.x == 0) spec_mpk[e * MPK_STRIDE + MPK_STRIDE - 1] = epoch;      // "this shadow belongs to step `epoch`"
        }
    }
}
// the games of `list` take their speculatively dealt successors: record (hot and cold part) and masks.  A game without a
// shadow of this step (the may-end filter of k_step is meant to be a superset; this is the safety net) is re-dealt here.
__global__ __launch_bounds__(64) void k_install_list(Ctx c, u32* __restrict__ mpk, Limits lim, const u32* __restrict__ count_p,
                                                     const i32* __restrict__ list, u8* __restrict__ busy, const u32* __restrict__ specR,
                                                     const u32* __restrict__ spec_mpk, u32 epoch, u32* __restrict__ err) {
    __shared__ __attribute__((aligned(16))) u32 rec[ROWS_HOT];
    __shared__ ResetScratch sc;
    const u32 count = *count_p;
    const int lane = threadIdx.x;
    for (u32 r = blockIdx.x; r < count; r += gridDim.x) {
        const long e = list[r];
        if (spec_mpk[e * MPK_STRIDE + MPK_STRIDE - 1] != epoch) {
            if (lane == 0) atomicAdd(err + 2, 1u);                                       // counted: catan_missed_speculation_count
            wave_reset_game(c, e, rec, sc, lane, mpk, lim, busy, nullptr);
            continue;
        }
        if (lane < REC / 4) reinterpret_cast<uint4*>(c.R + e * REC)[lane] = reinterpret_cast<const uint4*>(specR + e * REC)[lane];
        if (lane < 3)                                                                     // (words 0..11: the masks; the rest of the side row is the game's own)
            reinterpret_cast<uint4*>(mpk + e * MPK_STRIDE)[lane] = reinterpret_cast<const uint4*>(spec_mpk + e * MPK_STRIDE)[lane];
        if (lane == 0 && busy != nullptr) busy[e] = 0;
    }
}
// catan_reset: every game (sel == nullptr) or the selected ones.
__global__ __launch_bounds__(64) void k_reset(Ctx c, const u8* __restrict__ sel, int board_only) {
    __shared__ __attribute__((aligned(16))) u32 rec[ROWS_HOT];
    __shared__ ResetScratch sc;
    for (long e = blockIdx.x; e < c.N; e += gridDim.x) {
        if (sel != nullptr && (e >= c.n || sel[e] == 0)) continue;
        wave_reset_game(c, e, rec, sc, threadIdx.x, nullptr, Limits{ -1, -1 }, nullptr, nullptr, nullptr, nullptr, board_only != 0);
    }
}
// Contract (A): tops the two generators' output rings up to MT_AHEAD / MT_AHEAD_PY draws beyond the consumers' positions (the game's W_RNG,
// cons_py).  One lane: MT19937 is serial, and a handle under this contract holds ONE game (a step draws 0-3 words, a re-deal ~250).
__global__ __launch_bounds__(64) void k_mt_refill(Ctx c) {
    if (threadIdx.x != 0 || c.mt == nullptr) return;
    MtPair& m = *c.mt;
    const u32 want_np = c.R[W_RNG] + (u32)MT_AHEAD, want_py = m.cons_py + (u32)MT_AHEAD_PY;
    while ((int)(want_np - m.np.produced) > 0) { m.np.ring[m.np.produced & (u32)(MT_RING - 1)] = mt_next(m.np); m.np.produced++; }
    while ((int)(want_py - m.py.produced) > 0) { m.py.ring[m.py.produced & (u32)(MT_RING - 1)] = mt_next(m.py); m.py.produced++; }
}

// ------------------------------------------------------------------------------------------------ sort by action type
// Counting sort of the games by the type of the action they are about to take (18 bins: 13 types, play_dev split by card, no-op/padding), so
// that k_step's waves are type-homogeneous: the 13-way `switch` no longer serialises inside a wave.  The order inside
// a bin is irrelevant (games are independent), so block ranges are reserved with atomics.
DEVI int action_bin(const Ctx& c, const i32* __restrict__ actions, long e) {
    if (e >= c.n) return BIN_NOOP;
    return bin_of(actions[e * ACTION_WORDS], actions[e * ACTION_WORDS + 4]);
}
// Appends the block's games to the per-bin lists: rank inside the block from an LDS counter, the block's range in each
// bin reserved with one global atomic per (block, bin).  bins: this pass's NBINS counts (zero before the first block).
DEVI void sort_append(long e, bool valid, int bin, u32* hist, u32* base, u32* __restrict__ bins, i32* __restrict__ lists, long N) {
    u32 rank = 0;
    if (valid) rank = atomicAdd(&hist[bin], 1u);
    __syncthreads();
    if (threadIdx.x < NBINS) base[threadIdx.x] = hist[threadIdx.x] ? atomicAdd(&bins[threadIdx.x], hist[threadIdx.x]) : 0u;
    __syncthreads();
    if (valid) lists[(long)bin * N + base[bin] + rank] = (i32)e;
}
// the sort for caller-supplied actions (catan_step); the rollout loops do it inside k_sample_random
// err (validate mode, else null): an action type above 12 is an invalid action (no branch of validate_action matches,
// game/game.py:264-525) - counted here, the game then sits in the no-op bin; a NEGATIVE type is the C ABI's explicit no-op
// (a frozen game of a rollout), not an error.
__global__ __launch_bounds__(BLOCK) void k_classify(Ctx c, const i32* __restrict__ actions, u32* __restrict__ bins, i32* __restrict__ lists,
                                                    u32* __restrict__ zero_me, int zero_n, u32* __restrict__ err) {
    __shared__ u32 hist[NBINS], base[NBINS];
    if (threadIdx.x < NBINS) hist[threadIdx.x] = 0;
    __syncthreads();
    const long e = (long)blockIdx.x * BLOCK + threadIdx.x;
    if (zero_me != nullptr && e < zero_n) zero_me[e] = 0;
    if (err != nullptr && e < c.n && actions[e * ACTION_WORDS] > 12) atomicAdd(err, 1u);
    sort_append(e, e < c.n, e < c.n ? action_bin(c, actions, e) : BIN_NOOP, hist, base, bins, lists, c.N);
}
// ... and for caller-supplied actions under the deferred schedule (catan_step_deferred): a game still waiting for the slow path
// is in NO list (its action is ignored; k_step must not touch its result row, which a side stream may be writing).
__global__ __launch_bounds__(BLOCK) void k_classify_deferred(Ctx c, const i32* __restrict__ actions, const u8* __restrict__ busy, u32* __restrict__ bins,
                                                             i32* __restrict__ lists, u32* __restrict__ zero_me, int zero_n, u32* __restrict__ err) {
    __shared__ u32 hist[NBINS], base[NBINS];
    if (threadIdx.x < NBINS) hist[threadIdx.x] = 0;
    __syncthreads();
    const long e = (long)blockIdx.x * BLOCK + threadIdx.x;
    if (zero_me != nullptr && e < zero_n) zero_me[e] = 0;
    const bool plays = e < c.n && busy[e] == 0;
    if (err != nullptr && plays && actions[e * ACTION_WORDS] > 12) atomicAdd(err, 1u);
    sort_append(e, plays, plays ? action_bin(c, actions, e) : BIN_NOOP, hist, base, bins, lists, c.N);
}
// End of a catan_step_deferred call (the side work whose games return in the NEXT call has been joined): those games are
// released (tags tag_a / tag_b; release_all: catan_step_flush), and every game that is not waiting any more hands its step's
// result - written into the handle's result rows by whichever kernel completed the step - to the caller.
//   status 0: step complete (reward / done valid, state and masks current); 1: waiting (reward / done zero);
//   flush only: 2 = nothing was outstanding for this game (reward / done zero).
__global__ __launch_bounds__(BLOCK) void k_deliver(Ctx c, u8* __restrict__ busy, int tag_a, int tag_b, int release_all, const float* __restrict__ res_reward,
                                                   const u8* __restrict__ res_done, float* __restrict__ reward, u8* __restrict__ done, u8* __restrict__ status) {
    const long e = (long)blockIdx.x * BLOCK + threadIdx.x;
    if (e >= c.n) return;
    int b = busy[e];
    const bool was_waiting = b != 0;
    if (b != 0 && (release_all || (b >= 2 && (b == tag_a || b == tag_b)))) { busy[e] = 0; b = 0; }
    const bool deliver = b == 0 && (!release_all || was_waiting);
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    u8 d = 0;
    if (deliver) { r = *reinterpret_cast<const float4*>(res_reward + e * 4); d = res_done[e]; }
    *reinterpret_cast<float4*>(reward + e * 4) = r;
    done[e] = d;
    status[e] = b != 0 ? 1 : (deliver ? 0 : 2);
}
constexpr int SORT_PAD_WAVES = NBINS - 1;                // one partial wave per bin (the no-op bin comes last)

// ------------------------------------------------------------------------------------------------ random policy (kernel)
// bins != nullptr: also the histogram of the counting sort (k_classify_hist fused in; rollout loops) and the per-game
// action types for k_classify_scatter.  (Staging the mask / action rows through LDS with coalesced transfers was tried:
// 2.5 us slower - the kernel is bound by its divergent sampling chain, not by the row accesses.)
// BS threads per workgroup: 512 in the library's own deferred loop (43.2-43.4 against 43.5-43.6 us per pass with 256: half the range
// reservations per bin; 128 and 1 024 are slower: profiles/r05_s5_pass_experiments.txt, run 13)
template <int BS = BLOCK>
__global__ __launch_bounds__(BS) void k_sample_random(Ctx c, const u32* __restrict__ mpk, u32 step_idx, i32* __restrict__ actions,
                                                        u32* __restrict__ pctr, u8* __restrict__ busy, int tag_now, int tag_now2,
                                                        u32* __restrict__ zero_me, int zero_n, u32* __restrict__ bins, i32* __restrict__ lists) {
    __shared__ u32 hist[NBINS], base[NBINS];
    if (bins != nullptr) {
        if (threadIdx.x < NBINS) hist[threadIdx.x] = 0;
        __syncthreads();
    }
    St s(c.R, c.N, (long)blockIdx.x * BS + threadIdx.x);
    // the list counters this step / iteration starts from zero with: the tier-1 request counter of the iteration (deferred),
    // every slow-path list counter (lock-step: ctr[4..15]); nothing else touches them before k_step
    if (zero_me != nullptr && s.e < zero_n) zero_me[s.e] = 0;
    int t = BIN_NOOP;                                       // padding games: the no-op bin
    if (s.e < c.n) {
        u32 m[MASK_WORDS];
#pragma unroll
        for (int i = 0; i < MASK_WORDS; i++) m[i] = mpk[s.e * MPK_STRIDE + i];
        int a[ACTION_WORDS];
        t = sample_random(c, s, m, step_idx, a, pctr, busy, tag_now, tag_now2);
        t = bin_of(t, a[4]);
        uint2* row = reinterpret_cast<uint2*>(actions + s.e * ACTION_WORDS);         // 72 B rows: 8 B aligned
#pragma unroll
        for (int i = 0; i < ACTION_WORDS / 2; i++) row[i] = make_uint2((u32)a[2 * i], (u32)a[2 * i + 1]);
    }
    if (bins != nullptr) sort_append(s.e, s.e < c.n, t, hist, base, bins, lists, c.N);
}

// ------------------------------------------------------------------------------------------------ forward search
// Game.randomise_uncertainty (game/game.py:1207-1282): re-deal everything the controlling player cannot see.  The pile and
// the other players' hidden dev cards are pooled, shuffled and dealt back (popped from the right end, players in dict
// order Blue, Red, Orange, White); the other players' hands are reset to the controlling player's lower bounds and the
// unaccounted resource cards are handed out in a shuffled order, each to the first player of a freshly shuffled player
// order who is below his true hand size and below the controlling player's upper bound for that resource; the hand-out
// is retried until every resource adds up to 19 with the bank.  Lane = game; the per-game scratch arrays live in LDS as
// [index][lane].  ctrl[e] = controlling PlayerId 1..4 (0: leave game e alone).  The reference loops forever on states
// whose estimates admit no consistent deal; here the loop is capped (max_attempts) and such games are counted in err[1].
__global__ __launch_bounds__(64) void k_randomise_uncertainty(Ctx c, const i32* __restrict__ ctrl, u32* __restrict__ mpk, u32* __restrict__ err,
                                                              int max_attempts, Limits lim) {
    __shared__ u8 pool[32][64];
    __shared__ u8 lst[96][64];
    __shared__ u8 prop[20][64], hand[20][64], emx[20][64];
    const int lane = threadIdx.x;
    const long e = (long)blockIdx.x * 64 + lane;
    if (e >= c.n) return;
    const int cp = ctrl[e] - 1;
    if (cp < 0 || cp > 3) return;
    St s(c.R, c.N, e);
    Rng rng = rng_load(c, s);
    const int seatof = s.b(B_SEATOF);
    const u32 DICT = 1u | (3u << 8) | (2u << 16) | (0u << 24);           // Blue, Red, Orange, White as pid0 (game.py:18-23)
    // ---- development cards (:1210-1221)
    int n = 0;
    const int plen = s.b(B_PILE_LEN);
    for (int i = 0; i < plen; i++) pool[n++][lane] = (u8)s.pile(i);
    for (int k = 0; k < 4; k++) {
        const int p = (DICT >> (8 * k)) & 255;
        if (p == cp) continue;
        const int nh = s.pb(p, P_NHID);
        for (int j = 0; j < nh; j++) pool[n++][lane] = (u8)s.hidden(p, j);
    }
    for (int i = n - 1; i >= 1; i--) {                                   // np.random.shuffle
        const int j = (int)rng.bounded((u32)i);
        const u8 t = pool[i][lane]; pool[i][lane] = pool[j][lane]; pool[j][lane] = t;
    }
    for (int k = 0; k < 4; k++) {
        const int p = (DICT >> (8 * k)) & 255;
        if (p == cp) continue;
        const int nh = s.pb(p, P_NHID);
        u32 cnt = 0;                                                     // 5 counters of 6 bits
        for (int j = 0; j < nh; j++) { const int v = pool[--n][lane]; s.set_hidden(p, j, v); cnt += 1u << (6 * v); }
        for (int t = 0; t < 5; t++) s.spb(p, P_HCNT + t, (cnt >> (6 * t)) & 63);
    }
    s.sb(B_PILE_LEN, n);
    for (int i = 0; i < n; i++) s.set_pile(i, pool[i][lane]);
    // ---- resources (:1224-1243); r0: 0 Brick, 1 Wood, 2 Ore, 3 Sheep, 4 Wheat; reference order Sheep, Brick, Ore, Wheat, Wood
    const u32 RORD = 3u | (0u << 4) | (2u << 8) | (4u << 12) | (1u << 16);
    int total_before[4], unacc[5], bank[5];
#pragma unroll
    for (int r = 0; r < 5; r++) bank[r] = s.b(B_BANK + r);
#pragma unroll
    for (int p = 0; p < 4; p++) {
        int t = 0;
#pragma unroll
        for (int r = 0; r < 5; r++) { const int v = s.res(p, r); hand[p * 5 + r][lane] = (u8)v; t += v; }
        total_before[p] = t;
        if (p != cp) {
            Est E;
            est_load(s, cp, label_of(seatof, cp, p), E);
#pragma unroll
            for (int r = 0; r < 5; r++) { hand[p * 5 + r][lane] = (u8)E.mn[r]; emx[p * 5 + r][lane] = (u8)E.mx[r]; }
        }
    }
#pragma unroll
    for (int r = 0; r < 5; r++) {
        int acc = bank[r];
#pragma unroll
        for (int p = 0; p < 4; p++) acc += hand[p * 5 + r][lane];
        unacc[r] = 19 - acc;
    }
    int attempts = 0;
    bool ok = false;
    while (!ok) {                                                        // :1245-1276
        for (int x = 0; x < 20; x++) prop[x][lane] = hand[x][lane];
        int len = 0;
#pragma unroll
        for (int q = 0; q < 5; q++) {
            const int r = (RORD >> (4 * q)) & 15;
            for (int k = 0; k < unacc[r] && len < 96; k++) lst[len++][lane] = (u8)r;
        }
        for (int i = len - 1; i >= 1; i--) {                             // random.shuffle(res_list)
            const int j = (int)rng.bounded((u32)i);
            const u8 t = lst[i][lane]; lst[i][lane] = lst[j][lane]; lst[j][lane] = t;
        }
        while (len > 0) {
            const int r = lst[--len][lane];
            u32 keys = DICT;                                             // random.shuffle(player_keys)
#pragma unroll
            for (int i = 3; i >= 1; i--) {
                const int j = (int)rng.bounded((u32)i);
                const u32 a = (keys >> (8 * i)) & 255, b = (keys >> (8 * j)) & 255;
                keys = (keys & ~(255u << (8 * i)) & ~(255u << (8 * j))) | (b << (8 * i)) | (a << (8 * j));
            }
            for (int k = 0; k < 4; k++) {
                const int p = (keys >> (8 * k)) & 255;
                if (p == cp) continue;
                int tot = 0;
#pragma unroll
                for (int x = 0; x < 5; x++) tot += prop[p * 5 + x][lane];
                if (tot < total_before[p] && emx[p * 5 + r][lane] > prop[p * 5 + r][lane]) { prop[p * 5 + r][lane]++; break; }
            }
        }
        ok = true;
#pragma unroll
        for (int r = 0; r < 5; r++) {
            int in_hand = bank[r];
#pragma unroll
            for (int p = 0; p < 4; p++) in_hand += prop[p * 5 + r][lane];
            if (in_hand != 19) ok = false;
        }
        if (!ok && ++attempts >= max_attempts) { atomicAdd(&err[1], 1u); break; }
    }
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int r = 0; r < 5; r++) s.spb(p, P_RES + r, prop[p * 5 + r][lane]);
    s.sw(W_RNG, rng.draws);
    u32 m[MASK_WORDS];
    compute_masks(s, m, lim);
#pragma unroll
    for (int i = 0; i < MASK_WORDS; i++) mpk[e * MPK_STRIDE + i] = m[i];
}

// ------------------------------------------------------------------------------------------------ PMC calibration
// Streams n16 x 16 B from src to dst: a launch with exactly known HBM bytes, used to calibrate the rocprofv3
// FETCH_SIZE / WRITE_SIZE counters on this GPU (profiles/README.md).
__global__ __launch_bounds__(BLOCK) void k_calib_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, long n16) {
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n16; i += (long)gridDim.x * BLOCK) dst[i] = src[i];
}

// One wave that idles for `ticks` of the 100 MHz wall clock.  In front of tier 1 on its side stream it STAGGERS that launch behind the next pass's
// k_step (catan_abi.hip: T1_STAGGER_US); in front of a k_step (CATAN_DEBUG_STEP_DELAY_US, diagnostics) it makes that launch late, which turns a
// missing stream dependency on it into a deterministic failure (the fused loop's window close).
__global__ __launch_bounds__(64) void k_spin(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// ------------------------------------------------------------------------------------------------ deciding seat
// ref: env/wrapper.py:53-58, RL/ppo/game_manager.py:152-159.  out = PlayerId 1..4
// ignore_discard: the forward-search simulator's get_players_turn (RL/forward_search_policy/worker.py:146-151), which does
// not look at the discard phase
__global__ __launch_bounds__(BLOCK) void k_deciding(Ctx c, i32* __restrict__ out, int ignore_discard) {
    St s(c.R, c.N, (long)blockIdx.x * BLOCK + threadIdx.x);
    if (s.e >= c.n) return;
    int p;
    if (!ignore_discard && s.b(B_NDISC) > 0) p = s.b(B_DISC);
    else if (s.flags() & F_MUST_RESPOND) p = s.b(B_TRADE_TGT);
    else p = s.b(B_GO);
    out[s.e] = p + 1;
}

// ------------------------------------------------------------------------------------------------ export / import
// canonical int32 blob, layout spec.py STATE_FIELDS; blob is [STATE_WORDS][cnt] (word-major); idx = game ids or null
struct BlobW {
    i32* p; long cnt, i; int k;
    DEVI void put(int v) { p[(long)k * cnt + i] = v; k++; }
};
__global__ __launch_bounds__(BLOCK) void k_export(Ctx c, const long* __restrict__ idx, long cnt, i32* __restrict__ blob) {
    long i = (long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= cnt) return;
    St s(c.R, c.N, idx ? idx[i] : i);
    BlobW o{ blob, cnt, i, 0 };
    for (int t = 0; t < 19; t++) o.put(s.b(B_TILE + t) & 15);
    for (int t = 0; t < 19; t++) o.put(s.b(B_TILE + t) >> 4);
    o.put(s.b(B_ROBBER));
    for (int t = 0; t < 9; t++) o.put(s.b(B_HARB + t));
    u64 st[4], ct[4], rl[4]; u32 rh[4];
    for (int p = 0; p < 4; p++) { st[p] = s.settle(p); ct[p] = s.city(p); rl[p] = s.road_lo(p); rh[p] = s.road_hi(p); }
    for (int cn = 0; cn < 54; cn++) { int v = 0; for (int p = 0; p < 4; p++) { if ((st[p] >> cn) & 1) v = 1; if ((ct[p] >> cn) & 1) v = 2; } o.put(v); }
    for (int cn = 0; cn < 54; cn++) { int v = 0; for (int p = 0; p < 4; p++) if (((st[p] | ct[p]) >> cn) & 1) v = p + 1; o.put(v); }
    for (int e = 0; e < 72; e++) { int v = 0; for (int p = 0; p < 4; p++) if (e < 64 ? ((rl[p] >> e) & 1) : ((rh[p] >> (e - 64)) & 1)) v = p + 1; o.put(v); }
    for (int p = 0; p < 4; p++) {
        for (int r = 0; r < 5; r++) o.put(s.pb(p, P_RES + r));
        for (int r = 0; r < 5; r++) o.put(s.pb(p, P_VIS + r));
        Est E[3];
        for (int l = 0; l < 3; l++) est_load(s, p, l, E[l]);
        for (int l = 0; l < 3; l++) for (int r = 0; r < 5; r++) o.put(E[l].mn[r]);
        for (int l = 0; l < 3; l++) for (int r = 0; r < 5; r++) o.put(E[l].mx[r]);
        int hb = s.pb(p, P_HARB);
        for (int k = 0; k < 6; k++) o.put((hb >> k) & 1);
        int nh = s.pb(p, P_NHID);
        o.put(nh);
        for (int k = 0; k < 25; k++) o.put(k < nh ? s.hidden(p, k) : -1);
        int np = s.pb(p, P_NPLAYED);
        o.put(np);
        for (int k = 0; k < 25; k++) o.put(k < np ? s.played(p, k) : -1);
        o.put(s.pb(p, P_VP));
    }
    for (int r = 0; r < 5; r++) o.put(s.b(B_BANK + r));
    for (int p = 0; p < 4; p++) o.put(s.pb(p, P_SLEFT));
    for (int p = 0; p < 4; p++) o.put(s.pb(p, P_CLEFT));
    int pl = s.b(B_PILE_LEN);
    o.put(pl);
    for (int k = 0; k < 25; k++) o.put(k < pl ? s.pile(k) : -1);
    int order = s.b(B_ORDER);
    for (int k = 0; k < 4; k++) o.put(pid_at(order, k) + 1);
    o.put(s.b(B_ORDER_ID)); o.put(s.b(B_GO) + 1);
    int fl = s.flags();
    o.put((fl & F_INITIAL) ? 1 : 0);
    for (int p = 0; p < 4; p++) o.put(s.pb(p, P_ISET));
    for (int p = 0; p < 4; p++) o.put(s.pb(p, P_IROAD));
    for (int p = 0; p < 4; p++) { int v = s.pb(p, P_ISECOND); o.put(v == 255 ? -1 : v); }
    o.put((fl & F_ROLLED) ? 1 : 0); o.put((fl & F_PLAYED_DEV) ? 1 : 0); o.put((fl & F_MUST_USE_DEV) ? 1 : 0);
    int mr = (fl & F_MUST_RESPOND) ? 1 : 0;
    o.put(mr);
    o.put(mr ? s.b(B_TRADE_PROP) + 1 : 0); o.put(mr ? s.b(B_TRADE_TGT) + 1 : 0);
    o.put(s.b(B_TRADE_NG));
    for (int k = 0; k < 4; k++) o.put(s.b(B_TRADE_GIVE + k));
    o.put(s.b(B_TRADE_NR));
    for (int k = 0; k < 4; k++) o.put(s.b(B_TRADE_RECV + k));
    o.put((fl & F_RB_ACTIVE) ? 1 : 0); o.put(s.b(B_RB_COUNT));
    o.put((fl & F_CAN_ROBBER) ? 1 : 0); o.put((fl & F_JUST_ROBBER) ? 1 : 0);
    int nd = s.b(B_NDISC);
    o.put(nd > 0 ? 1 : 0); o.put(nd);
    for (int k = 0; k < 4; k++) o.put(k < nd ? s.b(B_DISC + k) + 1 : 0);
    o.put(s.b(B_DIE1)); o.put(s.b(B_DIE2));
    o.put(s.b(B_TRADES)); o.put((int)s.w(W_ACTIONS)); o.put((int)s.w(W_TURN));
    for (int k = 0; k < 5; k++) o.put(s.b(B_BOUGHT + k));
    o.put(s.b(B_LR_PLAYER)); o.put(s.b(B_LR_COUNT)); o.put(s.b(B_LA_PLAYER)); o.put(s.b(B_LA_COUNT));
    for (int p = 0; p < 4; p++) o.put(s.pb(p, P_CURLP));
    for (int p = 0; p < 4; p++) o.put(s.pb(p, P_ARMY));
    for (int p = 0; p < 4; p++) o.put(s.b(B_CURVP + p));
    o.put(s.b(B_WINNER));
    o.put((int)s.w(W_RNG));
}

struct BlobR {
    const i32* p; long cnt, i; int k;
    DEVI int get() { int v = p[(long)k * cnt + i]; k++; return v; }
};
__global__ __launch_bounds__(BLOCK) void k_import(Ctx c, const long* __restrict__ idx, long cnt, const i32* __restrict__ blob) {
    long i = (long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= cnt) return;
    St s(c.R, c.N, idx ? idx[i] : i);
    BlobR in{ blob, cnt, i, 0 };
    for (int r = 0; r < NROWS; r++) s.sw(r, 0);
    int tr[19];
    for (int t = 0; t < 19; t++) tr[t] = in.get();
    for (int t = 0; t < 19; t++) s.sb(B_TILE + t, tr[t] | (in.get() << 4));
    s.sb(B_ROBBER, in.get());
    for (int t = 0; t < 9; t++) s.sb(B_HARB + t, in.get());
    u64 bld1 = 0, bld2 = 0;
    for (int cn = 0; cn < 54; cn++) { int v = in.get(); if (v == 1) bld1 |= 1ull << cn; if (v == 2) bld2 |= 1ull << cn; }
    u64 st[4] = { 0, 0, 0, 0 }, ct[4] = { 0, 0, 0, 0 };
    for (int cn = 0; cn < 54; cn++) {
        int v = in.get();
        for (int p = 0; p < 4; p++) if (v == p + 1) { if ((bld1 >> cn) & 1) st[p] |= 1ull << cn; if ((bld2 >> cn) & 1) ct[p] |= 1ull << cn; }
    }
    for (int p = 0; p < 4; p++) { s.set_settle(p, st[p]); s.set_city(p, ct[p]); }
    u32 rd[4][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
    for (int e = 0; e < 72; e++) { int v = in.get(); for (int p = 0; p < 4; p++) if (v == p + 1) rd[p][e >> 5] |= 1u << (e & 31); }
    for (int p = 0; p < 4; p++) { s.sw(W_ROAD0 + p, rd[p][0]); s.sw(W_ROAD1 + p, rd[p][1]); s.sw(W_ROAD2 + p, rd[p][2]); }
    for (int p = 0; p < 4; p++) {
        for (int r = 0; r < 5; r++) s.spb(p, P_RES + r, in.get());
        for (int r = 0; r < 5; r++) s.spb(p, P_VIS + r, in.get());
        Est E[3];
        for (int l = 0; l < 3; l++) for (int r = 0; r < 5; r++) E[l].mn[r] = in.get();
        for (int l = 0; l < 3; l++) for (int r = 0; r < 5; r++) E[l].mx[r] = in.get();
        for (int l = 0; l < 3; l++) est_store(s, p, l, E[l]);
        int hb = 0;
        for (int k = 0; k < 6; k++) hb |= (in.get() ? 1 : 0) << k;
        s.spb(p, P_HARB, hb);
        int nh = in.get();
        s.spb(p, P_NHID, nh);
        int cnt5[5] = { 0, 0, 0, 0, 0 };
        for (int k = 0; k < 25; k++) {
            int v = in.get();
            if (k < nh) { s.set_hidden(p, k, v); for (int q = 0; q < 5; q++) cnt5[q] += (q == v) ? 1 : 0; }
        }
        for (int q = 0; q < 5; q++) s.spb(p, P_HCNT + q, cnt5[q]);
        int np = in.get();
        s.spb(p, P_NPLAYED, np);
        for (int k = 0; k < 25; k++) { int v = in.get(); if (k < np) s.set_played(p, k, v); }
        s.spb(p, P_VP, in.get());
    }
    for (int r = 0; r < 5; r++) s.sb(B_BANK + r, in.get());
    for (int p = 0; p < 4; p++) s.spb(p, P_SLEFT, in.get());
    for (int p = 0; p < 4; p++) s.spb(p, P_CLEFT, in.get());
    int pl = in.get();
    s.sb(B_PILE_LEN, pl);
    for (int k = 0; k < 25; k++) { int v = in.get(); if (k < pl) s.set_pile(k, v); }
    int order = 0, seatof = 0;
    for (int k = 0; k < 4; k++) { int p = in.get() - 1; order |= p << (2 * k); seatof |= k << (2 * p); }
    s.sb(B_ORDER, order); s.sb(B_SEATOF, seatof);
    s.sb(B_ORDER_ID, in.get()); s.sb(B_GO, in.get() - 1);
    int fl = 0;
    if (in.get()) fl |= F_INITIAL;
    for (int p = 0; p < 4; p++) s.spb(p, P_ISET, in.get());
    for (int p = 0; p < 4; p++) s.spb(p, P_IROAD, in.get());
    for (int p = 0; p < 4; p++) { int v = in.get(); s.spb(p, P_ISECOND, v < 0 ? 255 : v); }
    if (in.get()) fl |= F_ROLLED;
    if (in.get()) fl |= F_PLAYED_DEV;
    if (in.get()) fl |= F_MUST_USE_DEV;
    int mr = in.get();
    if (mr) fl |= F_MUST_RESPOND;
    { int a = in.get(), b = in.get(); s.sb(B_TRADE_PROP, mr ? a - 1 : 0); s.sb(B_TRADE_TGT, mr ? b - 1 : 0); }
    s.sb(B_TRADE_NG, in.get());
    for (int k = 0; k < 4; k++) s.sb(B_TRADE_GIVE + k, in.get());
    s.sb(B_TRADE_NR, in.get());
    for (int k = 0; k < 4; k++) s.sb(B_TRADE_RECV + k, in.get());
    if (in.get()) fl |= F_RB_ACTIVE;
    s.sb(B_RB_COUNT, in.get());
    if (in.get()) fl |= F_CAN_ROBBER;
    if (in.get()) fl |= F_JUST_ROBBER;
    s.sb(B_FLAGS, fl);
    in.get();                      // need_discard (derived from n_to_discard)
    int nd = in.get();
    s.sb(B_NDISC, nd);
    for (int k = 0; k < 4; k++) { int v = in.get(); s.sb(B_DISC + k, k < nd ? v - 1 : 0); }
    s.sb(B_DIE1, in.get()); s.sb(B_DIE2, in.get());
    s.sb(B_TRADES, in.get()); s.sw(W_ACTIONS, (u32)in.get()); s.sw(W_TURN, (u32)in.get());
    for (int k = 0; k < 5; k++) s.sb(B_BOUGHT + k, in.get());
    s.sb(B_LR_PLAYER, in.get()); s.sb(B_LR_COUNT, in.get()); s.sb(B_LA_PLAYER, in.get()); s.sb(B_LA_COUNT, in.get());
    for (int p = 0; p < 4; p++) s.spb(p, P_CURLP, in.get());
    for (int p = 0; p < 4; p++) s.spb(p, P_ARMY, in.get());
    for (int p = 0; p < 4; p++) s.sb(B_CURVP + p, in.get());
    s.sb(B_WINNER, in.get());
    { const u32 draws = (u32)in.get(); if (c.mt == nullptr) s.sw(W_RNG, draws); }    // (contract (A): the generators are the handle's, a restored game does not rewind them - as in the reference)
    LrCache lc;                    // nothing is known about the longest paths of an imported position
    lc.load(s.P);
    lc.invalidate_all();
    lc.store(s.P);
}

// test/diagnostic entry: longest path of player players[i] (PlayerId 1..4) in game i, unbudgeted tier-1 search
__global__ __launch_bounds__(64) void k_longest_path(Ctx c, const i32* __restrict__ players, i32* __restrict__ out) {
    __shared__ LrWave L;
    const int lane = threadIdx.x;
    const long e = (long)blockIdx.x * 64 + lane;
    St s(c.R, c.N, e);
    u32 nbr_c, nbr_e;
    lr_load_nbr(lane, nbr_c, nbr_e);
    const bool want = e < c.n;
    const int pid = want ? players[e] - 1 : 0;
    const int len = coop_longest_path(want, s, pid, L, 0, nbr_c, nbr_e);
    if (want) out[e] = len;
}

}  // namespace catan
