#!/bin/bash
# profiles/ from the outputs of tools/final_measure.sh (gpurun_out/final): run here after the GPU call
O=gpurun_out/final
cp $O/bench.json profiles/r1_bench_msm2p24.json
cp $O/bench_windowed.json profiles/r1_bench_msm2p24_windowed.json
python3 tools/prof_summary.py $O/prof_bench/bench_results.db 13 > profiles/r1_bench_msm2p24_kernel_stats.txt
python3 tools/pmc_summary.py $O/pmc_fetch/fetch_results.db $O/pmc_write/write_results.db $O/pmc_calib/calib_results.db 12 > profiles/r1_pmc_hbm_traffic.json
for f in msm_g1_sweep msm_g2_bls ntt msm_g1_table_sweep msm_g2_bls_table; do cp $O/$f.jsonl profiles/r1_$f.jsonl; done
cp $O/prove_joinsplit_shaped.json profiles/r1_prove_joinsplit_shaped.json
python3 tools/prof_summary.py $O/prof_prove/prove_results.db 12 > profiles/r1_prove_kernel_stats.txt
