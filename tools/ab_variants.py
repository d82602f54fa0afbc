"""Dev tool: time differently built render cores on the same resident scene.  python tools/ab_variants.py lib1.so lib2.so ..."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
workloads = os.environ.get("AB_WORKLOADS", "dragons_4k_ss4_d5,dragon_800_ss2_d5").split(",")
for lib in sys.argv[1:]:
    for wl in workloads:
        env = dict(os.environ, RTCORE_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--steps", "5", "--warmup", "2", "--no-cpu-baseline"],
                           env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(lib, wl, "FAILED", r.stderr[-500:], flush=True)
            continue
        j = json.loads(line[-1])
        cls = j["roofline"]["whole_frame"]["ms_by_kernel_class"]
        xd = j["parity"]["max_abs_diff_vs_reference_order_render"]
        print(f"{os.path.basename(lib):28s} {wl:22s} {j['ms_per_step']:8.2f} ms  {j['value']:8.1f} Mrays/s  " +
              "  ".join(f"{k} {v:.2f}" for k, v in cls.items()) + f"  xdiff {xd:.1e}", flush=True)
