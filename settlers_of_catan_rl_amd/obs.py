"""Observation encoder binding (filled in together with csrc/catan_obs.hip)."""


def single_env_obs(vec):
    return None
