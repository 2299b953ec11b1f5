"""Workload for SQ counter passes over k_tile_encoder_fwd (inference, 204 800 boards; REPS launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd import policy as P, nn_kernels
torch.manual_seed(0)
net = P.CatanPolicy().cuda()
te = net.observation_module.tile_encoder
boards = int(os.environ.get("BOARDS", "204800"))
tiles = (torch.rand((boards, 19, 60), device="cuda") < 0.2).to(torch.bfloat16)
with torch.no_grad():
    for _ in range(int(os.environ.get("REPS", "6"))):
        nn_kernels.tile_encoder_forward(te, tiles)
torch.cuda.synchronize()
