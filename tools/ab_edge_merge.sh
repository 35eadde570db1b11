# same-box A/B: the E-step's all-gather merged into the first objective round's (default build)
# against a build with -DNGHMM_NO_EDGE_MERGE (ngsf-hmm_amd/libnghmm_nomerge.so), one rank's compute
# of the 8- and 4-rank site-shard points with the RCCL all-gather in the loop
run() { python bench.py --steps 20 --warmup 4 --no_cpu_baseline --emulate_ranks $2 --emulate_rccl 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 V=$2', round(d['ms_per_step'],3), round(d['predicted']['all_gather_host_ms_per_call']*1e3,1), 'us/call')"; }
for rep in 1 2 3; do
  for v in 8 4; do
    NGHMM_LIB=$PWD/ngsf-hmm_amd/libnghmm_nomerge.so run "own gather " $v
    run "merged     " $v
  done
done
