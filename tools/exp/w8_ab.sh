# usage (GPU box): bash tools/exp/w8_ab.sh  -> parity suites on the W8 kernel, per-shape table and step A/B (w8 = 0 / 1)
python -m pytest tests/test_gpu_conv.py tests/test_gpu_split_bf16.py tests/test_gpu_winograd.py -x -q 2>&1 | tail -4
for w in 0 1; do echo "== w8=$w"; BENCH_TUNE=w8=$w python tools/bench_conv.py "res5 " 2>&1 | grep -v amdgpu; done
python tools/bench_winograd.py 2>&1 | tail -6
MRCNN_TUNE=w8=0 python tools/bench_winograd.py 2>&1 | tail -6
Q="--no-cpu-baseline --rotate-batches 0 --no-fg-capped --no-direct-head-forward --no-device-targets --pipeline-examples 0 --no-extra-workloads --no-fp32-mfma --repeats 3"
for w in 0 1 0 1; do echo "== step w8=$w"; python bench.py $Q --tune w8=$w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['repeats']['ms_per_step'], d['roofline']['achieved'], d['config']['loss'])"; done
