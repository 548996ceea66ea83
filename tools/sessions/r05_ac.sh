#!/bin/bash
# round 5 session ac: transposing split on an XCD-remapped 1-D grid (FF3D_SPLIT_ORDER=xcd) against the 3-D pixel-fastest grid: parity, A/B, fetch bytes
O=$PWD/gpurun_out/r05_ac; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
FF3D_SPLIT_ORDER=xcd timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_head_gpu.py -x -q -k "split or golden or conv" > $O/tests_xcd.log 2>&1; echo "rc=$?" >> $O/tests_xcd.log
tail -n 3 $O/tests_xcd.log | cut -c1-200
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
FF3D_SPLIT_ORDER=xcd b l_xcd
b l_pix
FF3D_SPLIT_ORDER=xcd b l_xcd2
b l_pix2
FF3D_SPLIT_ORDER=xcd b waymo_xcd --workload waymo --steps 10
b waymo_pix --workload waymo --steps 10
( cd /tmp && FF3D_SPLIT_ORDER=xcd timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc -o p -- python $R/bench.py --graph off --steps 3 --warmup 1 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/pmc.json 2> $O/pmc.err )
python tools/pmc_summary.py $(find $O/pmc -name '*_results.db' | head -1) split_nchw > $O/pmc_split_xcd_FETCH_SIZE.txt 2>&1
rm -rf $O/pmc
cat $O/pmc_split_xcd_FETCH_SIZE.txt | cut -c1-150
python - <<'PY'
import json
for n in ('l_xcd', 'l_pix', 'l_xcd2', 'l_pix2', 'waymo_xcd', 'waymo_pix'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_ac/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'))
    except Exception as e:
        print(n, 'no line', e)
PY
