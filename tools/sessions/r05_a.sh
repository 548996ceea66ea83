#!/bin/bash
# round 5 session a: torch-free graph/sync repro, the new round-5 tests, the 8-rank rehearsal, the default bench line (with `verified`)
mkdir -p gpurun_out/r05_a
O=gpurun_out/r05_a
hipcc --offload-arch=gfx950 -O2 tools/repro_graph_sync_fault.hip -o /tmp/repro_graph > $O/repro_build.log 2>&1
for v in control event_sync stream_sync device_sync pool; do
  timeout 60 /tmp/repro_graph $v > $O/repro_$v.log 2>&1; echo "variant $v rc=$?" >> $O/repro_summary.txt
done
cat $O/repro_summary.txt
timeout 1500 python -m pytest tests/test_round5_gpu.py -x -q > $O/tests_round5.log 2>&1; echo "rc=$?" >> $O/tests_round5.log
tail -5 $O/tests_round5.log
timeout 1500 python -m pytest tests/test_bench_cli_gpu.py -x -q > $O/tests_cli.log 2>&1; echo "rc=$?" >> $O/tests_cli.log
tail -5 $O/tests_cli.log
timeout 300 python -m pytest tests/test_small_batch_gpu.py -x -q -k "2-2" > $O/tests_small.log 2>&1; echo "rc=$?" >> $O/tests_small.log
tail -3 $O/tests_small.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads([l for l in open('gpurun_out/r05_a/bench_default.json') if l.startswith('{')][-1])
    print(d['value'], d['ms_per_step'], d['verified'], d['roofline']['frac'], d.get('configs3_strong', {}).get('value'),
          {k: v.get('value') for k, v in d.get('other_workloads', {}).items()}, d['cpu_baseline']['value'], d['cpu_baseline']['iqr'])
except Exception as e:
    print('no line', e)
PY
tail -5 $O/bench_default.err
