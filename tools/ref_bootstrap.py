"""Bootstrap for importing the upstream reference (development container ONLY).

The reference lives read-only at /root/reference and never travels to the GPU box.
`game/game.py:13` imports `ui.display` (pygame/tkinter, absent here), so a stub module
is pre-seeded before the import.  Nothing is copied from the
reference; this file only makes `from env.wrapper import EnvWrapper` work so that the
golden-vector generators under tools/ can drive it.
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


def bootstrap():
    if "ui.display" not in sys.modules:
        ui = types.ModuleType("ui")
        ui.__path__ = []
        disp = types.ModuleType("ui.display")
        disp.Display = type("Display", (), {})
        sys.modules["ui"] = ui
        sys.modules["ui.display"] = disp
    if "stable_baselines3" not in sys.modules:
        # vec_gather_experience.py:7 only needs CloudpickleWrapper (a .var holder).
        names = ["stable_baselines3", "stable_baselines3.common", "stable_baselines3.common.vec_env",
                 "stable_baselines3.common.vec_env.base_vec_env"]
        for n in names:
            m = types.ModuleType(n)
            m.__path__ = []
            sys.modules[n] = m

        class CloudpickleWrapper(object):
            def __init__(self, var):
                self.var = var

        sys.modules[names[-1]].CloudpickleWrapper = CloudpickleWrapper
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def have_reference():
    import os
    return os.path.isdir(REFERENCE_ROOT + "/game")
