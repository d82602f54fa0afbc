#!/bin/bash
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_mesh.py -m gpu -x -q --timeout 180 --timeout-method thread 2>&1 | tail -5 || exit 1
timeout -k 10 200 python tools/mesh_prep_bench.py animations/general/rotating-dragon 64 || exit 1
timeout -k 10 200 python tools/mesh_prep_bench.py images/general/dragons 16 || exit 1
