"""Dev tool: copy the summaries of gpurun_out/prof_<tag> (tools/profile.sh) into profiles/ and rebuild profiles/traffic.json.
usage: python tools/refresh_profiles.py r01h"""
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = f"gpurun_out/prof_{tag}"
shutil.copy(f"{src}/bench/bench_kernel_stats.csv", f"profiles/{tag}_bench_kernel_stats.csv")
line = [l for l in open(f"{src}/bench.log") if l.startswith("{")][-1]
open(f"profiles/{tag}_bench_line.json", "w").write(line)
for k in ("trace_shadow_kernel", "trace_closest_kernel", "shade_kernel", "combine_kernel"):
    json.dump(json.load(open(f"{src}/pmc_{k}.json")), open(f"profiles/{tag}_pmc_{k}_per_launch.json", "w"), indent=1)
j = json.load(open(f"profiles/{tag}_pmc_trace_shadow_kernel_per_launch.json"))
t = json.load(open("profiles/traffic.json"))
e = t["dragons_4k_ss4_d5"]
e["fetch_size_bytes_raw"] = j["FETCH_SIZE"] * 1024
e["write_size_bytes"] = j["WRITE_SIZE"] * 1024
e["hbm_bytes_per_launch"] = 2 * e["fetch_size_bytes_raw"] + e["write_size_bytes"]
e["profile"] = f"profiles/{tag}_pmc_trace_shadow_kernel_per_launch.json"
json.dump(t, open("profiles/traffic.json", "w"), indent=1)
b = json.loads(line)
r = b["roofline"]
print(f"{b['ms_per_step']:.2f} ms/frame, {b['value']:.0f} Mrays/s, frac {r['frac']:.3f}, kernel {r['kernel_ms']:.2f} ms/launch, "
      f"traffic {e['hbm_bytes_per_launch'] / 1e9:.2f} GB/launch")
print(r["whole_frame"]["ms_by_kernel_class"], r.get("fp64_vector", {}).get("achieved"))
print("lanes per VALU instr", j["SQ_THREAD_CYCLES_VALU"] / j["SQ_ACTIVE_INST_VALU"], "VALU busy", j["SQ_ACTIVE_INST_VALU"] / j["SQ_BUSY_CYCLES"] / 8,
      "wait", j["SQ_WAIT_ANY"] / j["SQ_WAVE_CYCLES"], "L2 hit", j["TCC_HIT_sum"] / j["TCC_REQ_sum"])
