"""Workload for the rocprofv3 --pmc passes (tools/profile_round.sh): the hot-path kernels at 65 536 games in the steady
state of the bench (the same kind of pre-roll: deferred passes until the games' ages are mixed), plus k_calib_copy launches
with exactly known HBM traffic (256 MiB read + 256 MiB written each).  tools/pmc_summarise.py averages the LAST launches of
every kernel, i.e. the measured section at the end, not the pre-roll."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd import _lib

CALIB_BYTES = 256 << 20
PREROLL = int(os.environ.get("PMC_PREROLL", "3072"))
env = VecCatanEnv(65536, seed=0)
L = _lib.lib()
a = torch.empty(CALIB_BYTES, dtype=torch.uint8, device="cuda").random_(0, 255)
b = torch.empty_like(a)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(4):
    _lib.check(L.catan_calib_copy(C.c_void_p(b.data_ptr()), C.c_void_p(a.data_ptr()), CALIB_BYTES, st))
torch.cuda.synchronize()
env.random_rollout_deferred(PREROLL, 32)         # pre-roll (its launches are profiled too, but not averaged)
env.random_rollout(1 << 20, 32)                  # lock-step passes: every kernel of the path, one after the other
env.random_rollout_deferred(96, 32)              # the measured section: the bench's schedule
torch.cuda.synchronize()
print("pmc workload done", int(env.policy_counters().sum()))
