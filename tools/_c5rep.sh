for k in 1 2 3 4 5 6 7 8; do
  NGHMM_DEBUG_CHECK=1 timeout -k 10 600 python bench.py --workload c5 --n_sites 200000 --gpus 2 --steps 2 --warmup 1 --no_cpu_baseline > gpurun_out/c5x.json 2> gpurun_out/c5x_$k.err
  python - "$k" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/c5x.json").read().strip().splitlines()[-1])
a = d["alt_sharding"]
print(sys.argv[1], d["check"]["rounds"], a.get("check", {}).get("rounds"), a.get("vs_main_sharding", {}).get("ok"), a.get("skipped"))
PY
done
for f in gpurun_out/c5x_*.err; do if grep -q "rounds 2 points" $f; then echo == $f; grep -h "\[loop\|\[check" $f | cut -c1-150; fi; done
