#!/bin/bash
# profiles/ from the outputs of tools/final_measure.sh (gpurun_out/final): run here after the GPU calls
O=gpurun_out/final
R=r3
cp $O/bench.json profiles/${R}_bench_msm2p24.json
cp $O/bench_windowed.json profiles/${R}_bench_msm2p24_windowed.json
python3 tools/prof_summary.py $O/prof_bench/bench_results.db 13 > profiles/${R}_bench_msm2p24_kernel_stats.txt
python3 tools/pmc_summary.py $O/pmc_fetch/fetch_results.db $O/pmc_write/write_results.db $O/pmc_calib/calib_results.db 12 > profiles/${R}_pmc_hbm_traffic.json
python3 tools/pmc_ntt_summary.py $O > profiles/${R}_pmc_ntt.json
for f in msm_g1_sweep msm_g2_bls ntt msm_g1_table_sweep msm_g2_bls_table; do cp $O/$f.jsonl profiles/${R}_$f.jsonl; done
cp $O/prove_joinsplit_shaped.json profiles/${R}_prove_joinsplit_shaped.json
cp $O/prove_bls377_2p22.json profiles/${R}_prove_bls377_2p22.json
python3 tools/prof_summary.py $O/prof_prove/prove_results.db 12 > profiles/${R}_prove_kernel_stats.txt
python3 tools/prof_timeline.py $O/prof_prove/prove_results.db > profiles/${R}_prove_timeline.txt
python3 tools/prof_overlap.py $O/prof_prove_if4/if4_results.db 0.84 0.98 > profiles/${R}_prove_overlap_4_in_flight.txt
{ echo "# rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace -- python3 tools/bench_prove.py --no-cpu --no-verify --in-flight 0 --steps 2"; echo "# (alt-bn128 JoinSplit-shaped proof, 2^19 domain; one proof of the run; tools/pmc_valu_by_kernel.py)"; python3 tools/pmc_valu_by_kernel.py $O/pmc_prove; } > profiles/${R}_prove_valu_by_kernel.txt
python3 tools/prof_proofs.py $O/prof_prove/prove_results.db > profiles/${R}_prove_per_proof.txt
