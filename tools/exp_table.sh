# scratch experiments for the fixed-base table mode (run on the GPU box)
cd $GRAFT_REPO_ROOT
P='import json,sys
for l in sys.stdin:
    j=json.loads(l); print(j["log_n"], j.get("table"), round(j["ms"],2), {k:round(v,2) for k,v in j["phases_ms"].items()}, j["verified"])'
timeout -k 10 300 python -u -m pytest tests/test_gpu_msm_table.py tests/test_gpu_msm.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2 &&
timeout -k 10 200 python tools/bench_sweep.py g1t 0 16 24 25 2>/dev/null | python3 -c "$P" &&
timeout -k 10 100 python tools/bench_sweep.py g1 2>/dev/null | head -2 | python3 -c "$P" &&
for T in 256 512 1024; do echo T=$T; ZG_MSM_SEGMENTS=$T timeout -k 10 200 python tools/bench_sweep.py g1t 0 21 24 2>/dev/null | python3 -c "$P" || exit 1; done
