"""Merges the two passes of tools/profile_policy_round.sh into profiles/r01_policy_mfma_utilisation.json:
MFMA FLOPs per launch = SQ_VALU_MFMA_BUSY_CYCLES (rocprofv3 --pmc pass, summed over the SIMDs) x 1024 FLOP per busy cycle
(dense bf16 MFMA rate of one SIMD: 2.5 PFLOP/s / 1024 SIMDs / 2.4 GHz, MI355X_MICROARCH.md); duration = AverageNs of the
separate --kernel-trace --stats pass.      python tools/mfma_utilisation.py gpurun_out/prof_policy_r01 out.json"""
import csv, json, sys

src, out = sys.argv[1], sys.argv[2]
summary = json.load(open(f"{src}/mfma_summary.json"))["kernels"]
stats = {}
for r in csv.DictReader(open(f"{src}/kernel_stats.csv")):
    stats[r["Name"][:90]] = r
rows = []
for k in summary:
    if k["mfma_busy_cycles_sum_over_simds"] <= 0 or k["kernel"] not in stats:
        continue
    st = stats[k["kernel"]]
    gflop = k["mfma_busy_cycles_sum_over_simds"] / k["launches"] * 1024 / 1e9
    avg_us = float(st["AverageNs"]) / 1e3
    rows.append({"kernel": k["kernel"], "calls_in_4_steps": int(st["Calls"]), "avg_us": avg_us, "total_ms": float(st["TotalDurationNs"]) / 1e6,
                 "mfma_gflop_per_launch": gflop, "achieved_tflops": gflop / avg_us * 1e3,
                 "pct_of_2500_tflops_bf16_peak": gflop / avg_us * 1e3 / 2500 * 100})
rows.sort(key=lambda r: -r["total_ms"])
json.dump({"workload": "4 training steps of the policy net at 65 536 rows, bf16 autocast (tools/pmc_policy_workload.py)",
           "method": "MFMA FLOPs per launch = SQ_VALU_MFMA_BUSY_CYCLES (rocprofv3 --pmc pass, summed over SIMDs) x 1024 FLOP per busy cycle "
                     "(bf16 MFMA at full rate: MI355X_MICROARCH.md); duration = AverageNs of the separate --kernel-trace --stats pass",
           "kernels": rows}, open(out, "w"), indent=1)
for r in rows[:14]:
    print(f"{r['kernel'][:64]:64s} {r['avg_us']:8.1f} us  {r['mfma_gflop_per_launch']:7.2f} GFLOP  {r['achieved_tflops']:7.1f} TFLOP/s  {r['pct_of_2500_tflops_bf16_peak']:5.2f} %")
