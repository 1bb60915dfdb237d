#!/bin/bash
# round 6, session zz (last rehearsal of the round, final tree): rehearsal of the driver's commands on a clean build: pytest -m gpu -x -q, smoke(), bench.py --gpus 1 --steps 20 --warmup 5
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06zz; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/ -x -q -m gpu > $O/tests.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt; tail -1 $O/smoke.log | cut -c1-300
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("it/s %.2f ms %.3f band2 %s" % (d["value"], d["ms_per_step"], d.get("value_band2")), d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), "traffic", d["roofline"]["traffic"], "build frac", round(d["roofline_build"]["frac"], 3), d["roofline_build"]["traffic"],
      "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"], "parity", d["cpu_baseline"]["parity_on_sample"]["sdf_max_rel_err"], d["cpu_baseline"]["parity_on_sample"]["rows_equal"])
PY
