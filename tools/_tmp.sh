timeout 600 python3 -m pytest tests/test_multigpu.py -x -q -m gpu -p no:cacheprovider -k "standalone_depthwise or other_configs" 2>&1 | tail -3
python bench.py > gpurun_out/r06s_bench.json 2> gpurun_out/r06s_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06s_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(json.dumps(d['standalone_depthwise'])[:3000])
PY
