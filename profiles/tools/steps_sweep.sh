#!/bin/bash
# GPU box: the default bench line at 10 / 20 / 40 / 100 timed steps (with the stage events of every fourth step, and
# without any), twice -- ms_per_step falls with the step count because the first ~13 steps after an idle stretch run on
# clocks that are still settling (profiles/tools/step_ramp.py shows every step).
for rep in 1 2; do
for k in 10 20 40 100; do
  timeout 200 python3 bench.py --no-cpu-baseline --no-extra-legs --steps $k --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($k, d['value'], d['ms_per_step'], d['roofline_encode']['ms'], d['roofline_decode']['ms'])"
  timeout 200 python3 bench.py --no-cpu-baseline --no-extra-legs --steps $k --warmup 5 --stage-events-every 1000 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($k, 'noev', d['value'], d['ms_per_step'])"
done; done
