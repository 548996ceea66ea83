#!/bin/bash
# round 3 session z: 8 x 32-pixel tile geometry of the halo conv (picked where it pads the map less: 468 x 468): tests under both
# forced geometries, A/B on the waymo workload and on the conv alone at 180 x 180 and 468 x 468
O=$PWD/gpurun_out/r03_z; mkdir -p $O
export TMPDIR=/tmp
for g in 0 1; do
  FF3D_HALO_GEO=$g timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_bench_shape_gpu.py -x -q -m gpu -k "conv or halo or head or dense or split" > $O/pytest_geo$g.log 2>&1; echo "tests (geometry $g) rc=$?"; tail -2 $O/pytest_geo$g.log | cut -c1-300
done
for g in 0 1 0 1; do echo -n "180x180 FF3D_HALO_GEO=$g: " | tee -a $O/halo_geo_ab.txt; FF3D_HALO_GEO=$g timeout 120 python tools/experiments/exp_halo.py 2>&1 | tail -1 | tee -a $O/halo_geo_ab.txt; done
show() { python - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]).read().strip().splitlines():
    if line.startswith('{'):
        d = json.loads(line)
        print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], {k: v for k, v in d['roofline_dense']['dense_launches_ms'].items() if k.startswith('conv3x3') and 's1' in k})
PY
}
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-strong-probe "$@" > $O/$name.json 2> $O/$name.err; show $O/$name.json; }
for rep in 1 2; do
b bench_waymo_auto_$rep --workload waymo
FF3D_HALO_GEO=0 b bench_waymo_geo0_$rep --workload waymo
done
b bench_b32_auto
FF3D_HALO_GEO=1 b bench_b32_geo1
