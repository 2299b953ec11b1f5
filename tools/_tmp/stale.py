import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd import _lib
CatanPolicy.refresh_kernel_packs = lambda self: None
import test_gpu_ppo_pipeline as t
try:
    t.test_collector_graphed_act_uses_current_weights(_lib.lib())
    print("NOT DETECTED")
except AssertionError as e:
    print("stale pack detected:", str(e)[:120])
