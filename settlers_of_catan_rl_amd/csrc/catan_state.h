// catan_state.h - packed per-game state records in HBM.
//
// Layout: one allocation per handle, game-major ("array of records"): game e owns the REC = 176 words
// R[e*REC .. e*REC+175] = 704 B = exactly 11 cache lines (64 B), 64 B aligned.
//     words 0 .. NW-1           : 32-bit fields (bitboards: corners 54 bit, edges 72 bit; packed estimates; counters)
//     bytes from word NW on     : byte fields, byte field b at byte offset 4*NW + b of the record
//     words 0 .. ROWS_HOT-1     : the HOT part = the first 7 cache lines (448 B): everything but the ordered dev-card
//                                 lists and the pile
// A record being contiguous is what lets a wave gather 64 ARBITRARY games at full line efficiency - k_step processes
// games sorted by action type (type-homogeneous waves) and stages their hot parts into an LDS tile transposed to
// tile[word][slot] (row stride 65 words: odd, so both the transposing writes and the lane-per-game reads are bank
// conflict free).  Lane-per-game kernels that touch little state (masks, sampler, export) address the records directly.
// 676 B of the 704 are used (the reference-like int32 form of the same state is 736 words = 2944 B, spec.py).
// The per-game side buffers are game-major too: actions int32 [n][18], rewards float [n][4], side rows u32 [N][32] (masks + next action).
//
// Players are indexed by pid0 = PlayerId-1 (0 White, 1 Blue, 2 Orange, 3 Red; reference game/enums.py:8-12).
// Resources r0 = Resource-1 (0 Brick, 1 Wood, 2 Ore, 3 Sheep, 4 Wheat; game/enums.py:22-28).
#pragma once
#include <stdint.h>

namespace catan {

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;

// ---- u32 rows
constexpr int W_SETTLE_LO = 0;    // +p : settlements of player p, corners 0..31
constexpr int W_SETTLE_HI = 4;    // +p : corners 32..53
constexpr int W_CITY_LO = 8;      // +p
constexpr int W_CITY_HI = 12;     // +p
constexpr int W_ROAD0 = 16;       // +p : edges 0..31
constexpr int W_ROAD1 = 20;       // +p : edges 32..63
constexpr int W_ROAD2 = 24;       // +p : edges 64..71
// opponent-hand estimates (Player.opponent_min_res / opponent_max_res, reference game/components/player.py:39-44):
// for observer o and label l (0 next, 1 next_next, 2 next_next_next) three words:
//   +0 : min of resources r0=0..3 (one byte each)   +1 : max of r0=0..3   +2 : min(r0=4) | max(r0=4) << 8
constexpr int W_EST = 28;         // + (o*3 + l)*3 + k   (36 rows)
constexpr int W_RNG = 64;         // philox draw counter of the game stream
constexpr int W_TURN = 65;
constexpr int W_ACTIONS = 66;     // actions_this_turn
constexpr int NW = 67;

// ---- byte fields, HOT part (staged in LDS by k_step)
constexpr int B_TILE = 0;         // +t (19): resource | value << 4
constexpr int B_HARB = 19;        // +slot (9): harbour id placed at slot
constexpr int B_ROBBER = 28;
constexpr int B_BANK = 29;        // +r0 (5)
constexpr int B_PILE_LEN = 34;
constexpr int B_ORDER = 35;       // player_order packed: seat i -> pid0 in bits 2i..2i+1
constexpr int B_SEATOF = 36;      // inverse: pid0 p -> seat in bits 2p..2p+1
constexpr int B_ORDER_ID = 37;
constexpr int B_GO = 38;          // players_go as pid0
constexpr int B_FLAGS = 39;
constexpr int B_RB_COUNT = 40;
constexpr int B_TRADE_PROP = 41;  // pid0
constexpr int B_TRADE_TGT = 42;   // pid0
constexpr int B_TRADE_NG = 43;
constexpr int B_TRADE_GIVE = 44;  // +i (4): Resource value 1..5
constexpr int B_TRADE_NR = 48;
constexpr int B_TRADE_RECV = 49;  // +i (4)
constexpr int B_NDISC = 53;       // len(players_to_discard); players_need_to_discard == (n > 0)
constexpr int B_DISC = 54;        // +i (4): pid0
constexpr int B_DIE1 = 58;
constexpr int B_DIE2 = 59;
constexpr int B_TRADES = 60;      // trades_proposed_this_turn
constexpr int B_LR_PLAYER = 61;   // 0 none, else PlayerId
constexpr int B_LR_COUNT = 62;
constexpr int B_LA_PLAYER = 63;   // 0 none, else PlayerId
constexpr int B_LA_COUNT = 64;
constexpr int B_BOUGHT = 65;      // +card (5): development_cards_bought_this_turn counts
constexpr int B_CURVP = 70;       // +p (4): EnvWrapper.curr_vps
constexpr int B_WINNER = 74;      // 0 none, else PlayerId
constexpr int B_PLAYER = 76;      // + p*PB + field
constexpr int PB = 26;
constexpr int P_RES = 0;          // +r0 (5)
constexpr int P_VIS = 5;          // +r0 (5)
constexpr int P_HARB = 10;        // bit 0 generic 3:1, bit r0+1 the 2:1 harbour of resource r0
constexpr int P_VP = 11;
constexpr int P_NHID = 12;
constexpr int P_NPLAYED = 13;
constexpr int P_HCNT = 14;        // +card (5): count of each card type in hidden_cards (derived)
constexpr int P_SLEFT = 19;
constexpr int P_CLEFT = 20;
constexpr int P_ISET = 21;
constexpr int P_IROAD = 22;
constexpr int P_ISECOND = 23;     // 255 = None
constexpr int P_CURLP = 24;       // current_longest_path
constexpr int P_ARMY = 25;        // current_army_size
constexpr int B_HOT = B_PLAYER + 4 * PB;    // 180 byte fields = 45 word rows
// ---- byte fields, COLD part (global only; touched by buy/play development card, reset, export)
constexpr int B_PILE = 180;       // +i (25): dev-card pile, popped from index pile_len-1
constexpr int B_CARDS = 208;      // + p*50 : hidden_cards[25] (ordered) then visible_cards[25] (ordered)
constexpr int NB = B_CARDS + 4 * 50;        // 408
static_assert(B_HOT % 4 == 0 && NB % 4 == 0, "byte fields are packed four per word row");

constexpr int ROWS_HOT = NW + B_HOT / 4;    // 112 words = 7 cache lines
constexpr int NROWS = NW + NB / 4;          // 169 words used
constexpr int REC = 176;                    // words per game record (704 B = 11 cache lines)
constexpr int ACTION_WORDS_C = 18;           // (= ACTION_WORDS below)
constexpr int TS = 65;                      // LDS tile row stride in words (odd: conflict-free transposition)
// Per-game SIDE ROW (128 B = one aligned L2 line, two HBM bursts): the packed masks and - inside the deferred rollouts, where the
// step samples the game's next action itself - that action and the game's decision counter, so that a stepping wave fetches
// one line per game next to the record and writes it back whole:
//   words 0..10 packed masks (325 bits)   11 zero   12..29 the next action (18 words)   30 decision counter   31 shadow tag
constexpr int MPK_STRIDE = 32;
constexpr int ROW_ACT = 12, ROW_CTR = 30, ROW_TAG = 31;
constexpr int STATE_BYTES_PER_GAME = REC * 4;
// ALGORITHMIC HBM bytes of one fused env step per stepped game (bench.py roofline): what an ideal k_step must move.
//   in : the action row (18 words), the HOT record (ROWS_HOT words)
//   out: the packed masks (11 words), reward float[4] + done byte, and the part of the record a step really changes -
//        the three counters (W_RNG, W_TURN, W_ACTIONS), the control block (robber .. winner: bytes B_ROBBER .. B_PLAYER-1),
//        two players' blocks (a trade / steal / monopoly moves cards between two hands), one (observer x label) set of
//        estimate words per observer of the acting player (3 x 3) and one 64-bit bitboard (a placement)
constexpr int WB_CONTROL_WORDS = (B_PLAYER - B_ROBBER + 3) / 4;
constexpr int WB_PLAYER_WORDS = (PB + 3) / 4;
constexpr int IDEAL_WRITEBACK_WORDS = 3 + WB_CONTROL_WORDS + 2 * WB_PLAYER_WORDS + 9 + 2;
constexpr int STEP_ALGO_BYTES = ACTION_WORDS_C * 4 + ROWS_HOT * 4 + 11 * 4 + 17 + IDEAL_WRITEBACK_WORDS * 4;
static_assert(WB_CONTROL_WORDS == 12 && WB_PLAYER_WORDS == 7 && IDEAL_WRITEBACK_WORDS == 40 && STEP_ALGO_BYTES == 741,
              "restate bench.py / DESIGN.md 6 when the layout changes (round 4: the step no longer reads its previous masks: 785 -> 741)");
// ... and of the FUSED-SAMPLING step (k_step<G, true>, the default deferred loop since round 6): the action and the game's decision counter come
// out of the game's side row and the next action goes back into it with the new masks (what the sampler kernel moved - masks in, action out -
// has moved into the step); the game's id is read from this pass's list and written to the next pass's
constexpr int STEP_FUSED_ALGO_BYTES = (ACTION_WORDS_C + 1) * 4 + ROWS_HOT * 4 + (11 + ACTION_WORDS_C + 1) * 4 + 17 + IDEAL_WRITEBACK_WORDS * 4 + 8;
static_assert(STEP_FUSED_ALGO_BYTES == 829, "restate bench.py / DESIGN.md 6 when the layout changes");
static_assert(NROWS * 4 == 676 && ROWS_HOT == 112 && ROWS_HOT * 4 % 64 == 0 && REC >= NROWS && REC * 4 % 64 == 0,
              "restate DESIGN.md byte table when the layout changes");

// B_FLAGS bits
constexpr int F_INITIAL = 1, F_ROLLED = 2, F_PLAYED_DEV = 4, F_MUST_USE_DEV = 8, F_MUST_RESPOND = 16,
              F_RB_ACTIVE = 32, F_CAN_ROBBER = 64, F_JUST_ROBBER = 128;

// enums (reference game/enums.py:30-50)
enum { C_KNIGHT = 0, C_VP = 1, C_YOP = 2, C_RB = 3, C_MONO = 4 };
enum { T_SETTLE = 0, T_ROAD = 1, T_CITY = 2, T_BUYDEV = 3, T_PLAYDEV = 4, T_EXCHANGE = 5, T_PROPOSE = 6,
       T_RESPOND = 7, T_ROBBER = 8, T_ROLL = 9, T_ENDTURN = 10, T_STEAL = 11, T_DISCARD = 12 };
// r0 indices
enum { R_BRICK = 0, R_WOOD = 1, R_ORE = 2, R_SHEEP = 3, R_WHEAT = 4 };

// flat mask offsets (EnvWrapper.get_action_masks, reference env/wrapper.py:172-185)
constexpr int M0 = 0, M1 = 13, M2 = 175, M3 = 248, M4 = 267, M5 = 272, M6 = 274, M7 = 283, M8 = 289, M9 = 295,
              M10 = 315, M11 = 320, MASK_BITS = 325, MASK_WORDS = 11;
constexpr int ACTION_WORDS = 18;
static_assert(ACTION_WORDS == ACTION_WORDS_C, "action words");
static_assert(ROW_ACT % 4 == 0 && ROW_ACT >= MASK_WORDS && ROW_ACT + ACTION_WORDS == ROW_CTR && ROW_TAG == MPK_STRIDE - 1, "side-row layout");
constexpr int STATE_WORDS = 736;   // canonical int32 blob (spec.py)
constexpr int OBS_FLOATS = 1787;

}  // namespace catan
