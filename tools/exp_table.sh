# scratch experiments (run on the GPU box)
cd $GRAFT_REPO_ROOT
P='import json,sys
for l in sys.stdin:
    j=json.loads(l); print(j["log_n"], (j.get("table") or {}).get("window_bits"), round(j["ms"],2), {k:round(v,2) for k,v in j["phases_ms"].items() if k.startswith("red")}, j["verified"])'
for B in 64 128; do for T in 1024 2048 4096; do echo B=$B T=$T; ZG_REDUCE_BLOCK=$B ZG_MSM_SEGMENTS=$T timeout -k 10 200 python tools/bench_sweep.py g1t 0 21 24 2>/dev/null | python3 -c "$P" || exit 1; done; done
