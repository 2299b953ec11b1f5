#!/bin/bash
# SURVEY section 5 / VERDICT r2 item 9: the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer.
# Builds oracle/libcatan_oracle_asan.so (`make -C oracle asan`: -fsanitize=address,undefined -fno-sanitize-recover=all, so a
# finding aborts the process) and drives it through (1) every oracle golden test (reference trajectories incl. the trade /
# action limits, reset states, longest-road cases, MT19937 KAT, GAE / PPO vectors) and (2) a random-policy fuzz: 2 048 games x
# 3 000 steps with auto-reset plus randomise_uncertainty calls along the way.  Usage: tools/fuzz_oracle_asan.sh [log]
set -eu
cd "$(dirname "$0")/.."
LOG=${1:-/dev/stdout}
make -C oracle -s asan
export CATAN_ORACLE_ASAN=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export OMP_NUM_THREADS=4
{
  echo "== oracle golden tests on the sanitized build"
  timeout 600 python -m pytest tests/test_oracle_golden.py -x -q -p no:cacheprovider 2>&1 | tail -3
  echo "== random-policy fuzz on the sanitized build"
  timeout 600 python - <<'PY'
import sys, time
sys.path.insert(0, "tests")
import numpy as np
import oracle_lib
assert oracle_lib.LIB_PATH.endswith("_asan.so")
t0 = time.time()
N = 2048
b = oracle_lib.OracleBatch(N, seed=20260928)
rng = np.random.default_rng(1)
steps = 0
for chunk in range(30):
    b.run_random(100, want_blobs=False, n_threads=4); steps += 100
    for i in rng.integers(0, N, size=64):              # Game.randomise_uncertainty on a sample of the games; the game then
        saved = np.zeros(oracle_lib.STATE_WORDS, dtype=np.int32)   # continues from the un-randomised state, as in the forward search
        p32 = oracle_lib.C.POINTER(oracle_lib.C.c_int32)           # (on a state that is not a true game state the reference's
        b.L.orc_export(b.env_ptr(int(i)), saved.ctypes.data_as(p32))   # rejection loop need not terminate)
        b.L.orc_randomise_uncertainty(b.env_ptr(int(i)), int(rng.integers(1, 5)))
        b.L.orc_import(b.env_ptr(int(i)), saved.ctypes.data_as(p32))
    m = b.masks()
    assert np.isfinite(m).all()
blobs = b.export()
print(f"fuzz ok: {N} games x {steps} steps, {b.games.value} games finished, 1920 randomise_uncertainty calls, "
      f"{time.time() - t0:.0f} s, no ASan/UBSan report (a report aborts the process)")
PY
} > "$LOG" 2>&1
