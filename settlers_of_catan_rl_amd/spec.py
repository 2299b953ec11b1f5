"""Canonical layouts shared by the host shim, the parity tests and the golden generators.

Three flat formats pin parity between the upstream reference (`/root/reference`, Python),
the CPU oracle (`oracle/catan_oracle.c`) and the HIP path (`csrc/catan_kernels.hip`):

* STATE BLOB  - int32[STATE_WORDS]: the full per-game state in reference-like, unpacked form
  (mirrors `Game.save_current_state`, reference game/game.py:1013-1091, plus the wrapper fields
  env/wrapper.py:30-34).  The device keeps a packed layout; `catan_state_export` unpacks to this.
* MASKS       - float32[325]: the 12 arrays of `EnvWrapper.get_action_masks`
  (env/wrapper.py:172-185) flattened row-major in head order.
* OBS         - float32[1787] + int32[5][OBS_LIST_PAD] card-id lists + lengths
  (env/wrapper.py:52-83, key order of RL/ppo/process_batch.py:10-13).
* ACTION      - int32[18]: the 12-head composite action of env/wrapper.py:114-166 flattened
  (heads 7 and 8 are 4-long sequences).
"""
from collections import OrderedDict

N_CORNERS, N_EDGES, N_TILES, N_PLAYERS = 54, 72, 19, 4
N_ACTION_TYPES = 13

# ---------------------------------------------------------------- action (int32[18])
ACTION_WORDS = 18
A_TYPE, A_CORNER, A_EDGE, A_TILE, A_CARD, A_RESPONSE, A_PLAYER = 0, 1, 2, 3, 4, 5, 6
A_GIVE, A_RECV, A_RES_A, A_RES_B, A_DISCARD = 7, 11, 15, 16, 17
# head index -> (offset, length) in the flat action
ACTION_HEAD_SLICES = [(0, 1), (1, 1), (2, 1), (3, 1), (4, 1), (5, 1), (6, 1), (7, 4), (11, 4), (15, 1), (16, 1), (17, 1)]

# ---------------------------------------------------------------- masks (float32[325])
MASK_SHAPES = [(13,), (3, 54), (73,), (19,), (5,), (2,), (3, 3), (6,), (6,), (4, 5), (5,), (5,)]
MASK_SIZES = []
for _s in MASK_SHAPES:
    _n = 1
    for _d in _s:
        _n *= _d
    MASK_SIZES.append(_n)
MASK_OFFSETS = [sum(MASK_SIZES[:i]) for i in range(len(MASK_SIZES))]
MASK_WORDS = sum(MASK_SIZES)
assert MASK_WORDS == 325
MASK_PACKED_WORDS = 11  # ceil(325 / 32) uint32 words, bit i of the flat mask -> word i>>5, bit i&31

# ---------------------------------------------------------------- obs
OBS_FLOAT_KEYS = OrderedDict([
    ("proposed_trade", (12,)),
    ("current_resources", (6,)),
    ("tile_representations", (19, 60)),
    ("current_player_main", (152,)),
    ("next_player_main", (159,)),
    ("next_next_player_main", (159,)),
    ("next_next_next_player_main", (159,)),
])
OBS_FLOAT_OFFSETS = OrderedDict()
_o = 0
for _k, _shape in OBS_FLOAT_KEYS.items():
    OBS_FLOAT_OFFSETS[_k] = _o
    _n = 1
    for _d in _shape:
        _n *= _d
    _o += _n
OBS_FLOATS = _o
assert OBS_FLOATS == 1787
OBS_LIST_KEYS = ["current_player_played_dev", "current_player_hidden_dev", "next_player_played_dev",
                 "next_next_player_played_dev", "next_next_next_player_played_dev"]
OBS_LIST_PAD = 25  # a player can hold at most the whole 25-card deck
# key order used by the reference rollout storage (RL/ppo/process_batch.py:10-13)
OBS_KEYS = ["proposed_trade", "current_resources", "tile_representations", "current_player_main",
            "current_player_played_dev", "current_player_hidden_dev", "next_player_main", "next_player_played_dev",
            "next_next_player_main", "next_next_player_played_dev", "next_next_next_player_main",
            "next_next_next_player_played_dev"]

# ---------------------------------------------------------------- state blob (int32)
def _player_fields(p):
    return [(f"p{p}_res", 5), (f"p{p}_vis", 5), (f"p{p}_opp_min", 15), (f"p{p}_opp_max", 15),
            (f"p{p}_harbours", 6), (f"p{p}_n_hidden", 1), (f"p{p}_hidden", 25), (f"p{p}_n_played", 1),
            (f"p{p}_played", 25), (f"p{p}_vp", 1)]


STATE_FIELDS = [("tile_res", 19), ("tile_val", 19), ("robber_tile", 1), ("harbour_type", 9),
                ("corner_bld", 54), ("corner_owner", 54), ("edge_owner", 72)]
for _p in (1, 2, 3, 4):
    STATE_FIELDS += _player_fields(_p)
STATE_FIELDS += [
    ("bank_res", 5), ("settlements_left", 4), ("cities_left", 4), ("pile_len", 1), ("pile", 25),
    ("player_order", 4), ("player_order_id", 1), ("players_go", 1),
    ("initial_phase", 1), ("init_settlements", 4), ("init_roads", 4), ("init_second_corner", 4),
    ("dice_rolled", 1), ("played_dev", 1), ("must_use_dev", 1), ("must_respond", 1),
    ("trade_proposer", 1), ("trade_target", 1), ("trade_n_give", 1), ("trade_give", 4),
    ("trade_n_recv", 1), ("trade_recv", 4),
    ("road_building_active", 1), ("road_building_count", 1),
    ("can_move_robber", 1), ("just_moved_robber", 1),
    ("need_discard", 1), ("n_to_discard", 1), ("to_discard", 4),
    ("die1", 1), ("die2", 1),
    ("trades_this_turn", 1), ("actions_this_turn", 1), ("turn", 1),
    ("bought_this_turn", 5),
    ("lr_player", 1), ("lr_count", 1), ("la_player", 1), ("la_count", 1),
    ("cur_longest_path", 4), ("cur_army_size", 4),
    ("curr_vps", 4), ("winner", 1),
    ("rng_draws", 1),
]
STATE_OFFSETS = OrderedDict()
_o = 0
for _name, _n in STATE_FIELDS:
    STATE_OFFSETS[_name] = (_o, _n)
    _o += _n
STATE_WORDS = _o


def state_field(blob, name):
    """View of one named field of a state blob (last axis = STATE_WORDS)."""
    off, n = STATE_OFFSETS[name]
    return blob[..., off:off + n]


def describe_state_diff(a, b, limit=12):
    """Human-readable list of differing fields between two blobs (for test failure messages)."""
    out = []
    for name, (off, n) in STATE_OFFSETS.items():
        xa, xb = a[off:off + n], b[off:off + n]
        if (xa != xb).any():
            out.append(f"{name}: {xa.tolist()} != {xb.tolist()}")
            if len(out) >= limit:
                break
    return "\n".join(out)
