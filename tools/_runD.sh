python tools/probes/pcie_pipeline_probe3.py 2>&1 | tail -6
for i in 1 2 3; do python bench.py --no-train --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['value_incl_pcie']; print(d['ms_per_step'], d['launch_mode']['timed_region'], v['in_flight'], round(v['one_at_a_time_ms_per_step'],3), round(v['pipelined_ms_per_step'],3), v.get('device_allocations_during'))"; done
