#!/bin/bash
# The measurements behind profiles/: run on the GPU box from the repository root, part 1 or 2.
#   part 1: GPU tests, smoke(), bench.py, kernel trace of bench.py, the two HBM counter passes (+ calibration)
#   part 3: counter passes over the transforms and over the accumulation kernel's address translation
#   part 2: size sweeps (MSM G1 2^16..2^26, G2 / BLS12-377 points, NTT), JoinSplit-shaped prove latency, throughput
#           with proofs in flight, and the prover's kernel trace
# Every GPU step runs under its own time limit and the chain stops at the first failure.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p "$O"
export TMPDIR=/tmp
cd "$R" || exit 1
if [ "$1" = "1" ]; then
    timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$O/gputests.log" 2>&1 &&
    tail -2 "$O/gputests.log" &&
    timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.log" 2>&1 &&
    tail -1 "$O/smoke.log" &&
    timeout -k 10 300 python bench.py > "$O/bench.json" 2> "$O/bench.err" &&
    cat "$O/bench.json" &&
    cd /tmp &&
    timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$O/prof_bench" -o bench -- python3 "$R/bench.py" --no-cpu-baseline --no-prove --no-windowed --no-g2 --no-ntt > "$O/prof_bench.log" 2>&1 &&
    timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O/pmc_fetch" -o fetch -- python3 "$R/bench.py" --no-cpu-baseline --no-prove --no-windowed --no-g2 --no-ntt --steps 3 --warmup 1 > "$O/pmc_fetch.log" 2>&1 &&
    timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$O/pmc_write" -o write -- python3 "$R/bench.py" --no-cpu-baseline --no-prove --no-windowed --no-g2 --no-ntt --steps 3 --warmup 1 > "$O/pmc_write.log" 2>&1 &&
    timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O/pmc_calib" -o calib -- "$R/build/calib_gather" > "$O/pmc_calib.log" 2>&1 &&
    for lg in 19 22; do
        timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_ntt${lg}_fetch" -- python3 "$R/tools/bench_sweep.py" ntt1 $lg 3 > "$O/pmc_ntt${lg}_fetch.log" 2>&1 &&
        timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_ntt${lg}_write" -- python3 "$R/tools/bench_sweep.py" ntt1 $lg 3 > "$O/pmc_ntt${lg}_write.log" 2>&1 || exit 1
    done &&
    echo part1 done
elif [ "$1" = "2" ]; then
    timeout -k 10 200 python tools/bench_sweep.py g1 > "$O/msm_g1_sweep.jsonl" 2> "$O/sweep.err" &&
    timeout -k 10 200 python tools/bench_sweep.py g1t 0 > "$O/msm_g1_table_sweep.jsonl" 2>> "$O/sweep.err" &&
    timeout -k 10 200 python tools/bench_sweep.py othert > "$O/msm_g2_bls_table.jsonl" 2>> "$O/sweep.err" &&
    timeout -k 10 100 python bench.py --no-table --no-prove --no-cpu-baseline > "$O/bench_windowed.json" 2>> "$O/sweep.err" &&
    timeout -k 10 200 python tools/bench_sweep.py other > "$O/msm_g2_bls.jsonl" 2>> "$O/sweep.err" &&
    timeout -k 10 100 python tools/bench_sweep.py ntt > "$O/ntt.jsonl" 2>> "$O/sweep.err" &&
    timeout -k 10 400 python tools/bench_prove.py --in-flight 4 --steps 40 --through-pool > "$O/prove_joinsplit_shaped.json" 2> "$O/prove.err" &&
    cat "$O/prove_joinsplit_shaped.json" &&
    cd /tmp &&
    timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$O/prof_prove" -o prove -- python3 "$R/tools/bench_prove.py" --no-cpu --in-flight 0 > "$O/prof_prove.log" 2>&1 &&
    timeout -s KILL 200 rocprofv3 --kernel-trace -d "$O/prof_prove_if4" -o if4 -- python3 "$R/tools/bench_prove.py" --no-cpu --no-verify --in-flight 4 --steps 20 > "$O/prof_prove_if4.log" 2>&1 &&
    timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$O/pmc_prove" -- python3 "$R/tools/bench_prove.py" --no-cpu --no-verify --in-flight 0 --steps 2 --warmup 1 > "$O/pmc_prove.log" 2>&1 &&
    cd "$R" && timeout -k 10 800 python tools/bench_prove.py --curve bls12-377 --constraints 4194302 --steps 12 --warmup 2 --in-flight 0 --no-verify > "$O/prove_bls377_2p22.json" 2> "$O/prove_bls377.err" &&
    echo part2 done
elif [ "$1" = "3" ]; then
    # counter passes: the transforms at 2^22 (tools/prof_ntt.sh) and the accumulation kernel's address translation
    bash tools/prof_ntt.sh 22 gpurun_out/final/prof_ntt22 &&
    cd /tmp &&
    timeout -s KILL 200 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum --kernel-trace --output-format csv -d "$O/pmc_utcl1" -- python3 "$R/bench.py" --no-cpu-baseline --no-prove --no-windowed --no-g2 --no-ntt --steps 3 --warmup 1 > "$O/pmc_utcl1.log" 2>&1 &&
    timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d "$O/pmc_tcc" -- python3 "$R/bench.py" --no-cpu-baseline --no-prove --no-windowed --no-g2 --no-ntt --steps 3 --warmup 1 > "$O/pmc_tcc.log" 2>&1 &&
    echo part3 done
else
    echo "usage: $0 1|2|3" >&2
    exit 2
fi
