# scratch experiments (run on the GPU box)
cd $GRAFT_REPO_ROOT
(while sleep 30; do date >> gpurun_out/tick.log; done) &
TICK=$!
trap "kill $TICK" EXIT
timeout -k 10 300 python -u -m pytest tests/test_gpu_prover.py tests/test_gpu_host_adapter.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2 &&
timeout -k 10 400 python tools/bench_prove.py --curve bls12-377 --constraints 4194302 --in-flight 0 > gpurun_out/prove_bls377_2p22.json 2> gpurun_out/prove_bls377_2p22.err &&
cut -c1-900 gpurun_out/prove_bls377_2p22.json &&
ZG_PK_TABLE_MIN=0 timeout -k 10 400 python tools/bench_prove.py --curve bls12-377 --constraints 4194302 --in-flight 0 --no-cpu --no-verify | cut -c1-400
