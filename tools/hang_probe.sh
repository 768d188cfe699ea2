#!/bin/bash
# Repeats the short reference-CI runs under a time limit and reports how each ended (round 6: one pytest run of
# tests/test_reference_ci_run.py stopped producing output; this separates the driver from the oracle half of the test).
N=${N:-20}
for i in $(seq $N); do
  s=$(date +%s.%N)
  timeout -k 5 60 ./examples/driver --yaml tests/golden/ci_input_pama.yaml --steps 5 - > /tmp/hp_$i.log 2>&1
  rc=$?
  e=$(date +%s.%N)
  echo "driver run $i rc=$rc $(echo "$e - $s" | bc) s $(tail -c 120 /tmp/hp_$i.log | tr '\n' ' ')" >> gpurun_out/hang_probe.log
  [ $rc -ne 0 ] && break
done
timeout -k 5 200 python -X faulthandler -m pytest tests/test_reference_ci_run.py -m gpu -q -x -k "first_crm_steps" >> gpurun_out/hang_probe.log 2>&1
echo "oracle half rc=$?" >> gpurun_out/hang_probe.log
