# scratch experiments for the fixed-base table mode (run on the GPU box)
cd $GRAFT_REPO_ROOT
P='import json,sys
for l in sys.stdin:
    j=json.loads(l); print(j["log_n"], j.get("table"), round(j["ms"],2), {k:round(v,2) for k,v in j["phases_ms"].items()}, j["verified"])'
for c in 20 21 22; do timeout -k 10 100 python tools/bench_sweep.py g1t $c 22 23 2>/dev/null | python3 -c "$P" || exit 1; done
echo L64; ZG_MSM_UNIT_LEN=64 timeout -k 10 100 python tools/bench_sweep.py g1t 21 22 2>/dev/null | python3 -c "$P" || exit 1
echo L512; ZG_MSM_UNIT_LEN=512 timeout -k 10 100 python tools/bench_sweep.py g1t 21 22 2>/dev/null | python3 -c "$P" || exit 1
echo L1024; ZG_MSM_UNIT_LEN=1024 timeout -k 10 100 python tools/bench_sweep.py g1t 21 22 2>/dev/null | python3 -c "$P" || exit 1
for c in 19 20 21; do timeout -k 10 100 python tools/bench_sweep.py g1t $c 21 2>/dev/null | python3 -c "$P" || exit 1; done
for c in 16 17 18 19; do timeout -k 10 100 python tools/bench_sweep.py g1t $c 19 20 2>/dev/null | python3 -c "$P" || exit 1; done
