"""Diagnostics: per-phase time of the fused k_step kernel at several game ages (run on the GPU box)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = VecCatanEnv(n, seed=0)
L = _lib.lib()
names = ["stage-in", "validate+apply (rest)", "LR request push", "(fine) switch body", "(fine) validate", None, "holder+done/reward+masks", "write-back"] if os.environ.get("CATAN_FINE_PROF") else ["stage-in", "validate+apply", "LR request push", None, None, None, "holder+done/reward+masks", "write-back"]
NP = len(names)
done = 0
for upto, chunk in [(128, 128), (3000, 256)]:
    env.random_rollout(done, upto - chunk - done); done = upto - chunk
    L.catan_profile_enable(env.h, 1)
    env.random_rollout(done, chunk); done = upto
    out = (C.c_uint64 * (2 * NP + 4 + 42))()
    L.catan_profile_read(env.h, out)
    L.catan_profile_enable(env.h, 0)
    waves = (n + 63) // 64
    print(f"--- steps {upto-chunk}..{upto}: mean us per wave-step | max us over all waves/steps (100 MHz ticks)")
    for i, nm in enumerate(names):
        if nm is None:
            continue
        print(f"  {nm:20s} mean {out[i] / (waves * chunk) / 100.0:9.2f} us   max {out[NP + i] / 100.0:9.2f} us")
    nres = max(1, out[4])
    print(f"  re-deals {out[4]} ({out[4]/chunk:.1f}/step): philox draws mean {out[3]/nres:.0f} max {out[NP+3]}; serial shuffle time mean {out[5]/nres/100.0:.1f} us max {out[NP+5]/100.0:.1f} us")
    print(f"  tier-1 requests {out[2*NP]} ({out[2*NP]/(waves*chunk):.2f}/wave-step), loop iterations {out[2*NP+1]} ({out[2*NP+1]/max(1,out[2*NP]):.1f}/request), overflows {out[2*NP+2]} ({out[2*NP+2]/chunk:.1f}/step)")
    tn = ["settle", "road", "city", "buy_dev", "play_dev", "exchange", "propose", "respond", "robber", "roll", "end_turn", "steal", "discard", "no-op"]
    base = 2 * NP + 4
    print("  validate+apply per action type (waves | mean us | max us): " + "  ".join(
        f"{tn[t]} {out[base+14+t]}|{out[base+t]/max(1,out[base+14+t])/100.0:.1f}|{out[base+28+t]/100.0:.1f}" for t in range(14) if out[base+14+t]))
