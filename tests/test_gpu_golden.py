"""GPU vs golden vectors captured from the upstream reference itself (tools/gen_golden.py): the HIP path is
checked directly against reference outputs, not only against the oracle.  Bit-exact."""
import numpy as np
import pytest

import golden_util as gu
from settlers_of_catan_rl_amd import spec

pytestmark = pytest.mark.gpu


def _env(n, seed, **kw):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    return VecCatanEnv(n, seed=seed, **kw)


def test_reset_states_vs_reference(hip_lib):
    g = gu.load("reset_states.npz")
    env = _env(len(g["blobs"]), int(g["seed"]))
    got = env.export_state().cpu().numpy()
    assert np.array_equal(got, g["blobs"]), spec.describe_state_diff(g["blobs"][0], got[0])


@pytest.mark.parametrize("name", gu.TRAJS)
def test_trajectory_vs_reference(hip_lib, name):
    import torch
    t = gu.load(name)
    env = _env(1, int(t["seed"]), env_id0=int(t["env_id"]), auto_reset=True)
    sample = {int(i): k for k, i in enumerate(t["sample_idx"])}
    n = len(t["actions"])
    for step in range(n):
        blob = env.export_state()[0].cpu().numpy()
        assert gu.crc(blob) == int(t["state_crc"][step]), f"state crc differs at step {step}"
        m = env.get_action_masks()[0].cpu().numpy()
        assert np.array_equal(m, gu.unpack_masks(t["masks"][step])), f"masks differ at step {step}"
        if step in sample:
            assert int(env.deciding_player()[0].item()) == int(t["deciding"][step])
            assert np.array_equal(blob, t["sample_blob"][sample[step]].astype(np.int32))
        a = torch.from_numpy(t["actions"][step].astype(np.int32)).view(1, spec.ACTION_WORDS)
        rew, done = env.step(a)
        assert np.array_equal(rew[0].cpu().numpy(), t["rewards"][step]) and bool(done[0].item()) == bool(t["dones"][step]), step
    assert env.invalid_action_count() == 0
    assert np.array_equal(env.export_state()[0].cpu().numpy(), t["final_blob"])


def test_reference_states_masks(hip_lib):
    """Import every sampled reference state of every golden trajectory into the device and compare the masks."""
    blobs, masks = [], []
    for name in gu.TRAJS:
        t = gu.load(name)
        for k, i in enumerate(t["sample_idx"]):
            blobs.append(t["sample_blob"][k].astype(np.int32)); masks.append(gu.unpack_masks(t["masks"][int(i)]))
    env = _env(len(blobs), 0)
    env.import_state(np.array(blobs))
    assert np.array_equal(env.export_state().cpu().numpy(), np.array(blobs))
    assert np.array_equal(env.get_action_masks().cpu().numpy(), np.array(masks))
