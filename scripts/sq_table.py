"""Per-kernel SQ counter table from scripts/gpu_pmc_sq.sh output (gpurun_out/pmc_sq/pmc_summary.json)."""
import json
import re
import sys

d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_sq/pmc_summary.json"))
tag = sys.argv[2] if len(sys.argv) > 2 else ""
print(f"rocprofv3 --pmc (scripts/gpu_pmc_sq.sh, bench.py --steps 1 --seconds 60{', ' + tag if tag else ''}): MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs);")
print("wave-time split = SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY, SQ_WAIT_ANY over SQ_WAVE_CYCLES; LDS = SQ_LDS_IDX_ACTIVE / (256 CUs x cycles) and the share of it that is SQ_LDS_BANK_CONFLICT")
print("kernel (grid = workgroups) | launches | MFMA busy | wave time: issuing | issue-stalled | waiting (waitcnt, barrier) | LDS active / CU cycle | conflict share of LDS cycles")
rows = []
for k, v in d.items():
    try:
        g = lambda n: v[n]["mean"]
        cyc = g("GRBM_GUI_ACTIVE") / 8
        wc = g("SQ_WAVE_CYCLES")
        rows.append((cyc * v["GRBM_GUI_ACTIVE"]["launches"], k, v["GRBM_GUI_ACTIVE"]["launches"], g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * cyc),
                     g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_WAIT_ANY") / wc, g("SQ_LDS_IDX_ACTIVE") / (256 * cyc),
                     g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1)))
    except KeyError:
        continue
rows.sort(reverse=True)
for r in rows[:28]:
    name = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", r[1])
    print("%-58s | %4d | %5.1f %% | %5.1f %% | %5.1f %% | %5.1f %% | %4.2f | %4.2f" % (name[:58], r[2], 100 * r[3], 100 * r[4], 100 * r[5], 100 * r[6], r[7], r[8]))
