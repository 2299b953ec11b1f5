"""bench.py prints its ONE JSON line whatever happens: a failure before or after the timed region, or a SIGTERM from
the launcher while the main thread sits in a blocking call, yields the line with what was measured so far and a `status`."""
import json
import os
import signal
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_line_with_status_when_the_run_fails_early():
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["status"].startswith("failed during stage 'start'") and "HIP device" in line["status"], line
    assert line["value"] is None and line["metric"].startswith("Catan env-steps/sec")


def test_line_on_sigterm_while_the_main_thread_blocks():
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "bench.PARTIAL.update(value=123.0, n_gpus=8); bench.stage('ppo_update'); bench._watch_sigterm()\n"
            "print('ready', flush=True)\n"
            "time.sleep(120)\n") % ROOT
    env = dict(os.environ, RANK="0")
    p = subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.stdout.readline().decode().strip() == "ready"
    time.sleep(0.2)
    p.send_signal(signal.SIGTERM)
    out, _ = p.communicate(timeout=60)
    assert p.returncode == 1
    line = json.loads(out.decode().strip().splitlines()[-1])
    assert line["value"] == 123.0 and "SIGTERM" in line["status"] and "ppo_update" in line["status"], line
