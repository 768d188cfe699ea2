#!/bin/bash
# run the CI config N times per variant and print the distinct (substeps, temp_min, temp_max, maxw) outcomes with their counts
N=${N:-14}
for v in "$@"; do
  echo "== variant [$v]"
  for i in $(seq $N); do ./examples/driver --yaml tests/golden/ci_input_pama.yaml $v - 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['substeps'], d['temp_min'], d['temp_max'], d['maxw_final'])"; done | sort | uniq -c
done
