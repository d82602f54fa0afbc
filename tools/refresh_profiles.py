"""Dev tool: copy the summaries of gpurun_out/prof_<tag> (tools/profile.sh) into profiles/ and rebuild profiles/traffic.json.
usage: python tools/refresh_profiles.py r02a"""
import json
import os
import shutil
import sys

tag = sys.argv[1]
wl = sys.argv[2] if len(sys.argv) > 2 else "dragons_4k_ss4_d5"
src = f"gpurun_out/prof_{tag}"
shutil.copy(f"{src}/bench/bench_kernel_stats.csv", f"profiles/{tag}_bench_kernel_stats.csv")
line = [l for l in open(f"{src}/bench.log") if l.startswith("{")][-1]
open(f"profiles/{tag}_bench_line.json", "w").write(line)
t = {wl: {}}
for k in ("trace_shadow_kernel", "trace_closest_kernel", "shade_kernel", "combine_kernel"):
    j = json.load(open(f"{src}/pmc_{k}.json"))
    json.dump(j, open(f"profiles/{tag}_pmc_{k}_per_launch.json", "w"), indent=1)
    fetch, write = j["FETCH_SIZE"] * 1024, j["WRITE_SIZE"] * 1024
    t[wl][k] = {
        "fetch_size_bytes_raw": fetch, "write_size_bytes": write,
        "hbm_bytes_per_launch": 2 * fetch + write,
        "l2_request_bytes_per_launch": j["TCC_REQ_sum"] * 128,
        "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_REQ_sum in separate passes (tools/profile.sh), mean per launch of the "
               "kernel; KB -> bytes x1024; read side doubled per MI355X_MICROARCH.md HBM section (gfx950 FETCH_SIZE tallies 128-B "
               "requests at 64 B; calibrated there for wide streaming reads, so for this gather pattern the read side lies between "
               "fetch_size_bytes_raw and twice that); L2: TCC_REQ_sum x 128 B",
        "profile": f"profiles/{tag}_pmc_{k}_per_launch.json"}
json.dump(t, open("profiles/traffic.json", "w"), indent=1)
b = json.loads(line)
r = b["roofline"]
print(f"{b['ms_per_step']:.2f} ms/frame, {b['value']:.0f} Mrays/s, frac {r['frac']:.3f}, kernel {r['kernel_ms']:.2f} ms/launch")
print(r["whole_frame"]["ms_by_kernel_class"])
for k in ("trace_shadow_kernel", "trace_closest_kernel"):
    j = json.load(open(f"profiles/{tag}_pmc_{k}_per_launch.json"))
    print(k, "lanes per VALU instr", j["SQ_THREAD_CYCLES_VALU"] / j["SQ_ACTIVE_INST_VALU"], "VALU busy", j["SQ_ACTIVE_INST_VALU"] / j["SQ_BUSY_CYCLES"] / 8,
          "wait", j["SQ_WAIT_ANY"] / j["SQ_WAVE_CYCLES"], "L2 hit", j["TCC_HIT_sum"] / j["TCC_REQ_sum"], "VALU instr", j["SQ_INSTS_VALU"])
