#!/bin/bash
# parity tests that exercise the trace kernels, then the bench line
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread 2>&1 | tail -5 || exit 1
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/quick_bench.log 2>&1 || { tail -5 gpurun_out/quick_bench.log; exit 1; }
python - <<'PY'
import json
j=[json.loads(l) for l in open("gpurun_out/quick_bench.log") if l.startswith("{")][-1]
print(j["ms_per_step"], j["value"], j["roofline"]["whole_frame"]["ms_by_kernel_class"], j["parity"], j["roofline"]["frac"])
PY
